// msc_pair_gemm.hip -- the Q x M pass on the matrix cores (gfx950 MFMA), r04 form: ONE int8 matrix product per tile of bins over ONE
// BIT per bin, exact for ANY counts.
//
// The pass needs three integer reductions per (query q, candidate c) (pair_features.hip): sum |q_i - c_i| (manhattan, intersection,
// kulczynski2: predict/Feature.cpp:859-871,764-777,682-695), sum q_i c_i (euclidean, normalized_vectors, pearson, simratio:
// :1113-1124,1171-1184,795-811,829-841) and the earth mover's distance (:1505-1518; msc_emd_ranks.hip). With excess counts
// e = count - 1 (every bin starts at the pseudocount 1, nonltr/KmerHashTable.cpp:69-72):
//     sum |q_i - c_i| = sum e_q + sum e_c - 2 sum min(e_q, e_c)          sum q_i c_i = 4^k + sum e_q + sum e_c + sum e_q e_c
// An L-base sequence touches at most L of the 4^k bins, almost all of them ONCE (a 1 kb sequence at k = 9 has ~2 bins with e >= 2), and
// where both excesses are 0 or 1, min(e_q, e_c) = e_q e_c = f_q f_c with the presence bits f = [e >= 1]. Write e = f + g (g = e - 1
// on the "large bins" e >= 2, else 0). Then, exactly:
//     P1(q, c) = sum_i f_q(i) f_c(i)        the number of shared k-mers: one v_mfma_i32_32x32x32_i8 per 32 queries x 32 candidates x 32 bins
//     sum min(e_q, e_c) = P1 + sum over bins large in BOTH of [ min(e_q, e_c) - 1 ]
//     sum e_q e_c       = P1 + sum_i g_q(i) f_c(i)  +  sum over the candidate's large bins of g_c e_q
//                              ^ P2: the queries' large bins      ^ walked by the epilogue
// The corrections run over SHORT LISTS, not over bins: per slot the (bin, e) pairs with e >= 2 (~2 per 1 kb sequence at k = 9, one more per
// repeat unit), kept beside the mirror. The candidate's list is walked by the epilogue (k_pair_epilogue_bits in pair_features.hip: per entry
// one coalesced read of the queries' counts at that bin); the queries' lists become a per-step "hot list" this kernel consults while the
// candidates' bits of that step sit in its registers (P2: one atomic add per (entry, candidate that holds the k-mer) -- 0.4 % of the
// candidates at k = 9). Exact in integers for any counts of the narrow range (<= 8191): the r03 form (thermometer levels over a byte per
// bin) ran only while the LARGEST count of both whole sets was <= 16 -- one homopolymer run among 100 000 sequences sent every pair to the
// digest kernel at a fifth of the rate -- and streamed 8 x the bytes.
//
//   kb mirror         msc_kbits.h: a bit per bin, slots blocked by 32; + the lists of large bins (mb, pitch entries per slot, unordered)
//   k_kb_gather       the queries' side of a block of <= QN queries: their bits in the order the GEMM stages them (16 bytes per row and
//                     128-bin step) and their counts transposed [bin][query] (bytes, 0 / 1 here) for the epilogue's lookups
//   k_hot_*           the queries' large bins bucketed by 128-bin step: (bin, query row, e - 1); the fill also writes the counts of those
//                     bins into the transposed image
//   k_pair_gemm_bits  workgroup = 128 candidates x QN queries x one slice of the bins; wave = 32 candidates x QN queries: QN / 32
//                     accumulators of 32 x 32. A lane loads 16 bytes of its candidate per 256 bins and expands 16 bits to the 16 bytes of
//                     a B operand with 12 VALU operations (v_bfe, v_mul_u32_u24, v_and per dword), shared by the QN / 32 products of
//                     that k-chunk; the queries' tile of a 128-bin step is expanded once per workgroup into LDS (16-byte segments
//                     XOR-swizzled so that a 32-row A operand read is conflict-free). Per 32 x 32 x 32 tile: one ds_read_b128 + one MFMA.
//                     Roofline: the int8 matrix pipe (5 POPS dense) -- 4^k multiply-adds per pair; HBM sees 4^k / 8 bytes per candidate
//                     and QN queries.
//   output            int32 P1 [slice][candidate][QN] (plain stores, the epilogue adds the slices), int32 P2 [candidate][QN] (atomics)
#include "msc_internal.h"
#include "msc_wave.h"
#include "msc_kbits.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr uint32_t kStep = 128;          // bins per step of the queries' tile; the candidates' bits are loaded per 256 (two steps)

// 16 presence bits -> the 16 bytes (0 / 1) of an MFMA operand: per dword one nibble, spread by a multiply (n * 0x204081 puts bit i of
// the nibble at bit 8 i: the four shifted copies do not overlap) -- v_bfe_u32, v_mul_u32_u24, v_and_b32
__device__ __forceinline__ v4i expand16(uint32_t hw) {
	v4i r;
#pragma unroll
	for (int d = 0; d < 4; d++) r[d] = (int)(__umul24((hw >> (4 * d)) & 0xfu, 0x00204081u) & 0x01010101u);
	return r;
}

// ------------------------------------------------------------------------------------------------ the mirror and its lists
// 16 bins per thread = one halfword of the mirror. flags[0] |= 1 when a zero count is met (the identities above need every count >= 1:
// KmerHashTable's initial value, and a mean of such histograms too); flags[1] = the longest list seen (the host re-lays the lists out
// when it passes the pitch).
template <typename T>
__global__ void __launch_bounds__(256) k_kb_build(const T* __restrict__ bins, uint64_t nbins, uint8_t* __restrict__ kb, uint64_t first_slot, uint64_t n_slots,
                                                  uint2* __restrict__ mb, uint32_t* __restrict__ mb_n, uint32_t pitch, int32_t* __restrict__ flags) {
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (i >= n_slots * nbins) return;
	const uint64_t slot = first_slot + i / nbins, at = i % nbins;
	constexpr int NV = sizeof(T);          // 16-byte loads for 16 bins
	uint32_t raw[4 * NV];
	const uint4* src = reinterpret_cast<const uint4*>(bins + slot * nbins + at);
#pragma unroll
	for (int v = 0; v < NV; v++) { const uint4 q = src[v]; raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w; }
	uint32_t hw = 0;
	bool zero = false;
#pragma unroll
	for (int j = 0; j < 16; j++) {
		uint32_t c;
		if constexpr (sizeof(T) == 1) c = (raw[j >> 2] >> (8 * (j & 3))) & 0xffu;
		else if constexpr (sizeof(T) == 2) c = (raw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
		else c = raw[j];
		if (c == 0) { zero = true; continue; }
		const uint32_t e = c - 1;
		if (e) hw |= 1u << j;
		if (e >= 2) {
			const uint32_t pos = atomicAdd(&mb_n[slot], 1u);
			if (pos < pitch) mb[slot * pitch + pos] = make_uint2((uint32_t)at + j, e);
			atomicMax(&flags[1], (int32_t)(pos + 1));
		}
	}
	*reinterpret_cast<uint16_t*>(kb + msc_kb_offset(slot, at, nbins)) = (uint16_t)hw;
	if (zero) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------------ the queries' side of a block
// One workgroup per 128-bin step, one thread per query row. abits[step][row] = 8 halfwords, halfword 2 kc + h = the bits of bins
// 32 kc + 16 h .. + 15 of the step (what lane half h feeds k-chunk kc); rows past n_q are zero. qT[bin][row] = 0 / 1 (k_hot_fill then
// writes the counts of the large bins over it).
// fp4_image: the order k_pair_gemm_fp4 stages instead -- abits[super-step][row][segment 2 t + h] = dword t of half h of the row's 32 mirror
// bytes of the super-step (what lane half h feeds the super-step's product t).
template <int QN>
__global__ void __launch_bounds__(256) k_kb_gather(const uint8_t* __restrict__ kb, const uint32_t* __restrict__ q_slots, uint32_t n_q, uint64_t nbins,
                                                   uint8_t* __restrict__ abits, uint8_t* __restrict__ qT, int fp4_image, uint8_t* __restrict__ anib) {
	__shared__ uint32_t qbits[kStep * (QN / 32)];          // [bin of the step][word of 32 rows]
	const uint32_t step = blockIdx.x, row = threadIdx.x;
	uint32_t hwv_keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (row < QN) {
		uint32_t hwv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		if (row < n_q) {
			// the mirror keeps a 256-bin super-step as [h][8 halfwords j]: this step is j = 4 (step & 1) .. + 3 of both halves
			const uint8_t* src = kb + msc_kb_offset(q_slots[row], (uint64_t)(step >> 1) * 256, nbins) + 8 * (step & 1);
			const v2i lo = *reinterpret_cast<const v2i*>(src), hi = *reinterpret_cast<const v2i*>(src + 16);
#pragma unroll
			for (int kc = 0; kc < 4; kc++) {
				hwv[2 * kc] = ((uint32_t)lo[kc >> 1] >> (16 * (kc & 1))) & 0xffffu;
				hwv[2 * kc + 1] = ((uint32_t)hi[kc >> 1] >> (16 * (kc & 1))) & 0xffffu;
			}
		}
		v4i packed;
#pragma unroll
		for (int d = 0; d < 4; d++) packed[d] = (int)(hwv[2 * d] | (hwv[2 * d + 1] << 16));
		if (fp4_image) *reinterpret_cast<v4i*>(abits + (((uint64_t)(step >> 1) * QN + row) * 8 + 4 * (step & 1)) * 4) = v4i{(int)(hwv[0] | (hwv[2] << 16)), (int)(hwv[1] | (hwv[3] << 16)), (int)(hwv[4] | (hwv[6] << 16)), (int)(hwv[5] | (hwv[7] << 16))};
		if (anib) {          // the queries' tile as k_pair_gemm_fp4_dma copies it into LDS: nibbles, segments already swizzled
			v4i* tile_row = reinterpret_cast<v4i*>(anib) + ((uint64_t)(step >> 1) * QN + row) * 8;
#pragma unroll
			for (uint32_t x = 0; x < 4; x++) {          // x = 2 (t & 1) + h of this step's two products
				const uint32_t h = x & 1, t = 2 * (step & 1) + (x >> 1), w = hwv[4 * (x >> 1) + h] | (hwv[4 * (x >> 1) + 2 + h] << 16);
				tile_row[(2 * t + h) ^ ((row >> 1) & 7)] = v4i{(int)((w << 2) & 0x44444444u), (int)(w & 0x22222222u), (int)((w >> 2) & 0x11111111u), (int)((w >> 1) & 0x44444444u)};
			}
		}
		else *reinterpret_cast<v4i*>(abits + ((uint64_t)step * QN + row) * 16) = packed;
#pragma unroll
		for (int sg = 0; sg < 8; sg++) hwv_keep[sg] = hwv[sg];
	}
	// transposed, as BITS: bin p of the step (halfword p / 16 of the row, bit p % 16) -> one ballot per bin and wave of 64 rows; lane l of a
	// wave keeps the ballots of bins l and l + 64. Record of a bin = [plane][QN / 32 words]: plane 0 = "the query holds this k-mer"
	// (e >= 1), plane 1 = "... more than once" (e >= 2: set by k_hot_fill; the epilogue then reads the count in the query's own list).
	if (row < ((QN + 63u) & ~63u)) {          // (whole waves: a lane past QN holds zeros and keeps the ballots of its two bins like any other)
		const uint32_t lane = row & 63, wv = row >> 6;
		uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll
		for (uint32_t pth = 0; pth < kStep; pth++) {
			const unsigned long long b = __builtin_amdgcn_ballot_w64((hwv_keep[pth >> 4] >> (pth & 15)) & 1u);
			// (a select, not v_writelane from inline asm: the compiler pads no wait states between the compare that writes the ballot's
			// scalar registers and an asm statement that reads them, and the lane then keeps the PREVIOUS bin's ballot now and then)
			const bool mine = lane == (pth & 63);
			lo[pth >> 6] = mine ? (uint32_t)b : lo[pth >> 6];
			hi[pth >> 6] = mine ? (uint32_t)(b >> 32) : hi[pth >> 6];
		}
#pragma unroll
		for (uint32_t hf = 0; hf < 2; hf++) {
			qbits[(64 * hf + lane) * (QN / 32) + 2 * wv] = lo[hf];
			if (QN >= 64) qbits[(64 * hf + lane) * (QN / 32) + 2 * wv + 1] = hi[hf];
		}
	}
	__syncthreads();
	// one 16-byte piece per thread: (bin, plane, piece) in the record's own order; plane 1 leaves here as zeros
	constexpr uint32_t PPB = QN / 128 ? QN / 128 : 1;          // 16-byte pieces per plane of a record (QN < 128: the record is padded to 16 bytes per plane)
	for (uint32_t it = threadIdx.x; it < kStep * 2 * PPB; it += 256) {
		const uint32_t pbin = it / (2 * PPB), plane = (it / PPB) & 1, piece = it % PPB;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (plane == 0) {
			const uint32_t* src = qbits + pbin * (QN / 32) + 4 * piece;
			v.x = src[0];
			if (QN >= 64) v.y = src[1];
			if (QN >= 128) { v.z = src[2]; v.w = src[3]; }
		}
		*reinterpret_cast<uint4*>(qT + (((uint64_t)step * kStep + pbin) * 2 * PPB + plane * PPB + piece) * 16) = v;
	}
}

// The queries' large bins, bucketed by step: count, exclusive scan, fill. An entry = (bin, row << 16 | e - 1); the fill also puts
// min(e, 127) at [bin][row] of the transposed image (behind k_kb_gather on the stream).
__global__ void __launch_bounds__(256) k_hot_count(const uint2* __restrict__ mb, const uint32_t* __restrict__ mb_n, uint32_t pitch, const uint32_t* __restrict__ q_slots,
                                                   uint32_t n_q, uint32_t* __restrict__ cnt) {
	const uint32_t row = blockIdx.x;
	if (row >= n_q) return;
	const uint32_t slot = q_slots[row];
	const uint32_t n = mb_n[slot] < pitch ? mb_n[slot] : pitch;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[mb[(uint64_t)slot * pitch + i].x / kStep], 1u);
}
__global__ void __launch_bounds__(1024) k_hot_scan(const uint32_t* __restrict__ cnt, uint32_t nsteps, uint32_t* __restrict__ ptr, uint32_t* __restrict__ cursor) {
	__shared__ uint32_t part[1024];
	const uint32_t per = (nsteps + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nsteps ? lo + per : nsteps;
	uint32_t s = 0;
	for (uint32_t i = lo; i < hi; i++) s += cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = part[threadIdx.x] - s;
	for (uint32_t i = lo; i < hi; i++) { ptr[i] = run; cursor[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) ptr[nsteps] = part[1023];
}
__global__ void __launch_bounds__(256) k_hot_fill(const uint2* __restrict__ mb, const uint32_t* __restrict__ mb_n, uint32_t pitch, const uint32_t* __restrict__ q_slots,
                                                  uint32_t n_q, uint32_t qn, uint32_t* __restrict__ cursor, uint2* __restrict__ hot, uint8_t* __restrict__ qT) {
	const uint32_t row = blockIdx.x;
	if (row >= n_q) return;
	const uint32_t slot = q_slots[row];
	const uint32_t n = mb_n[slot] < pitch ? mb_n[slot] : pitch;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint2 en = mb[(uint64_t)slot * pitch + i];
		const uint32_t g = en.y - 1 > 0xffffu ? 0xffffu : en.y - 1;          // (counts of the narrow range are <= 8191)
		hot[atomicAdd(&cursor[en.x / kStep], 1u)] = make_uint2(en.x, (row << 16) | g);
		atomicOr(reinterpret_cast<uint32_t*>(qT) + msc_qt_word(en.x, 1, row, qn), 1u << (row & 31));
	}
}

// ------------------------------------------------------------------------------------------------ the product
// NRB = QN / 32 row blocks of queries. Registers: 16 NRB accumulators + 8 of candidate bits + the operands in flight: 3 waves per SIMD up
// to QN = 128, 2 at 256.
template <int NRB, int NW>
__global__ void __launch_bounds__(64 * NW, NRB == 8 ? 2 : (NW == 8 ? 2 : 3)) k_pair_gemm_bits(const uint8_t* __restrict__ cand_kb, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                                          const uint8_t* __restrict__ abits, uint64_t nbins, uint32_t k_slices, const uint32_t* __restrict__ hot_ptr,
                                                                          const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff) {
	constexpr int QN = 32 * NRB;
	constexpr int NT = 64 * NW;              // NW waves of 32 candidates each share the queries' tile
	constexpr int SPT = QN * 8 / NT;         // 16-byte segments of the tile a thread expands per step
	constexpr int TPR = 8 / SPT;             // threads that stage one query row
	static_assert(SPT >= 1 && SPT <= 8 && TPR * SPT == 8, "tile staging");
	__shared__ v4i sA[2][QN * 8];          // [buffer][row][16-byte segment ^ ((row >> 1) & 7)]: QN x 128 bytes each
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices, k0 = (uint64_t)ks * per;
	const uint32_t steps = (uint32_t)(per / kStep), gstep0 = (uint32_t)(k0 / kStep), n_ss = steps / 2;          // (the host makes `steps` even)
	const uint32_t ci = (blockIdx.x * NW + wave) * 32 + (lane & 31);
	const bool valid = ci < m;
	const uint32_t cc = valid ? ci : m - 1;
	const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
	// lane l: candidate l % 32 of the block, half l / 32: its 16 bytes of every 256-bin super-step; consecutive slots of one block make
	// the wave's load one contiguous KiB
	const uint8_t* brow = cand_kb + (slot >> 5) * msc_kb_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16 + (k0 >> 8) * 1024;
	// staging: thread -> (row, NRB segments). Rows are dealt so that the 8 lanes of a ds_write_b128 group write 8 different swizzled
	// segments (rows 0, 2, .. 14, then 1, 3, .. 15 of each 16)
	const uint32_t u = tid / TPR, part = tid % TPR;
	const uint32_t srow = (u & ~15u) | (2 * (u & 7) + ((u >> 3) & 1));
	const uint8_t* asrc = abits + (uint64_t)srow * 16 + part * (2 * SPT);
	v16i acc[NRB];
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int i = 0; i < 16; i++) acc[rb][i] = 0;
	uint32_t a_reg[(SPT + 1) / 2];          // SPT halfwords of the row's bits
	auto fetch_a = [&](uint32_t i) {
		const uint8_t* p = asrc + (uint64_t)(gstep0 + (i < steps ? i : steps - 1)) * (QN * 16);
		if constexpr (SPT == 8) { const v4i v = *reinterpret_cast<const v4i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; a_reg[2] = v.z; a_reg[3] = v.w; }
		else if constexpr (SPT == 4) { const v2i v = *reinterpret_cast<const v2i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; }
		else if constexpr (SPT == 2) a_reg[0] = *reinterpret_cast<const uint32_t*>(p);
		else a_reg[0] = *reinterpret_cast<const uint16_t*>(p);
	};
	auto park = [&](uint32_t buf) {
#pragma unroll
		for (int t = 0; t < SPT; t++) {
			const uint32_t sg = part * SPT + t;
			sA[buf][srow * 8 + (sg ^ ((srow >> 1) & 7))] = expand16((a_reg[t >> 1] >> (16 * (t & 1))) & 0xffffu);
		}
	};
	auto fetch_b = [&](uint32_t ss) { return *reinterpret_cast<const v4i*>(brow + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * 1024); };
	// one 128-bin step: k-chunks kc = 0 .. 3 = the four halfwords of (w0, w1)
	auto multiply = [&](uint32_t buf, uint32_t w0, uint32_t w1) {
		const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
		for (int kc = 0; kc < 4; kc++) {
			const v4i B = expand16(((kc < 2 ? w0 : w1) >> (16 * (kc & 1))) & 0xffffu);
#pragma unroll
			for (int rb = 0; rb < NRB; rb++) {
				// A operand: lane l = query 32 rb + l % 32, the bins of half l / 32 of this 32-bin chunk
				const v4i A = sA[buf][(32 * rb + r) * 8 + ((2 * kc + hh) ^ sw)];
				acc[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[rb], 0, 0, 0);
			}
		}
	};
	// P2: the queries' large bins that fall into this step, while the candidates' bits of the step are in registers. Wave-uniform
	// entries; the lane that holds (its candidate, that bin) adds e_q - 1 for its pair when the bit is set (0.4 % of them at k = 9).
	auto hotfix = [&](uint32_t i, const v4i& bq) {
		const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[gstep0 + i]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[gstep0 + i + 1]);
		for (uint32_t e = h0; e < h1; e++) {
			const uint2 en = hot[e];
			const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
			const uint32_t j = (bin >> 5) & 7, wsel = j >> 1;
			const uint32_t w = (uint32_t)(wsel == 0 ? bq.x : wsel == 1 ? bq.y : wsel == 2 ? bq.z : bq.w);
			const uint32_t bit = (w >> (16 * (j & 1) + (bin & 15))) & 1u;
			if (valid && (lane >> 5) == ((bin >> 4) & 1) && bit) atomicAdd(out_diff + (uint64_t)ci * QN + (rg >> 16), (int32_t)(rg & 0xffffu));
		}
	};
	v4i bcur = fetch_b(0);
	fetch_a(0);
	park(0);
	__syncthreads();
	for (uint32_t ss = 0; ss < n_ss; ss++) {          // a super-step of 256 bins = two steps of the queries' tile; the LDS halves swap roles by name
		const v4i bnext = fetch_b(ss + 1);            // (the last one once more past the end: a load nobody uses is cheaper than a branch around it)
		fetch_a(2 * ss + 1);
		multiply(0, (uint32_t)bcur.x, (uint32_t)bcur.y);
		if (hot_ptr) hotfix(2 * ss, bcur);
		park(1);
		__syncthreads();
		fetch_a(2 * ss + 2);
		multiply(1, (uint32_t)bcur.z, (uint32_t)bcur.w);
		if (hot_ptr) hotfix(2 * ss + 1, bcur);
		park(0);
		__syncthreads();
		bcur = bnext;
	}
	// D: lane l holds column l % 32 (its candidate); register 4 g + j = row 8 g + 4 (l / 32) + j of each 32-query block
	if (!valid) return;
	int32_t* o = out_min + ((uint64_t)ks * m + ci) * QN + 4 * (lane >> 5);
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int g = 0; g < 4; g++) *reinterpret_cast<v4i*>(o + 32 * rb + 8 * g) = v4i{acc[rb][4 * g], acc[rb][4 * g + 1], acc[rb][4 * g + 2], acc[rb][4 * g + 3]};
}

// ------------------------------------------------------------------------------------------------ the product, FP4 form
// The same product on v_mfma_f32_32x32x64_f8f6f4 with both operands in FP4 (E2M1): 64 bins per instruction in the cycles the int8 form takes
// for 32 (MI355X: ~10 POPS dense against ~5), half the LDS bytes per bin, and an expansion of bits that costs a v_and per EIGHT bins:
// the product only needs both operands to agree on which bin sits in which K position, so nibble i of operand dword d takes bit 4 i + d of
// the 32-bit word -- a mask. The nibble then holds 1, 2, 4 (or, shifted down, 1): as E2M1 0.5, 1.0, 2.0. The candidates' side uses
// (0.5, 1, 2, 0.5) for d = 0 .. 3 and the queries' side (2, 1, 0.5, 2), so every product of two set bits is exactly 1.0; the sums (at most
// the number of k-mers of a sequence) are exact in the f32 accumulators and are converted to int32 on the way out. (tools/probes/
// fp4_mfma_probe.hip checks the identity D[i][j] = popcount(a_i & b_j) on the card; the non-scaled opcode -- both scale operands the
// constant 0 -- scales by 1.) One tile step = 256 bins = one 16-byte load of a candidate = four products per row block.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr uint32_t kM1 = 0x11111111u, kM2 = 0x22222222u, kM4 = 0x44444444u;

// QS = 1: a wave owns 32 candidates and all NRB row blocks (one LDS read of a queries' operand per product). QS = 2: the workgroup's tile
// is dealt the other way -- a wave owns 64 candidates (two column tiles) and NRB / 2 row blocks, so that one LDS read serves TWO products
// (the CU's LDS delivers 128 bytes per clock: exactly the operand bytes the matrix pipe consumes at its peak when every product reads its
// own 16 bytes per lane); the price is that two waves expand the same candidate bits.
template <int NRB, int NW, int QS>
__global__ void __launch_bounds__(64 * NW, NRB == 8 ? 2 : 4) k_pair_gemm_fp4(const uint8_t* __restrict__ cand_kb, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                                     const uint8_t* __restrict__ abits, uint64_t nbins, uint32_t k_slices, const uint32_t* __restrict__ hot_ptr,
                                                                     const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff) {
	constexpr int QN = 32 * NRB;
	constexpr int NT = 64 * NW;
	constexpr int SPT = QN * 8 / NT;         // 16-byte segments of the tile (= dwords of the bit image) a thread expands per step
	constexpr int TPR = 8 / SPT;
	constexpr int CT = QS;                   // column tiles (of 32 candidates) per wave
	constexpr int RBW = NRB / QS;            // row blocks per wave
	static_assert(SPT >= 1 && SPT <= 8 && TPR * SPT == 8, "tile staging");
	static_assert(NRB % QS == 0 && NW % QS == 0, "tile deal");
	__shared__ v4i sA[2][QN * 8];          // [buffer][row][segment (2 t + h) ^ ((row >> 1) & 7)]: QN x 128 bytes = 256 bins of nibbles
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t qg = wave % QS, cg = wave / QS;
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices, k0 = (uint64_t)ks * per;
	const uint32_t n_ss = (uint32_t)(per / 256), gss0 = (uint32_t)(k0 / 256);
	uint32_t ci[CT];
	bool valid[CT];
	const uint8_t* brow[CT];
#pragma unroll
	for (int c = 0; c < CT; c++) {
		ci[c] = (blockIdx.x * NW + cg * CT + c) * 32 + (lane & 31);
		valid[c] = ci[c] < m;
		const uint32_t cc = valid[c] ? ci[c] : m - 1;
		const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
		brow[c] = cand_kb + (slot >> 5) * msc_kb_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16 + (k0 >> 8) * 1024;
	}
	const uint32_t u = tid / TPR, part = tid % TPR;
	const uint32_t srow = (u & ~15u) | (2 * (u & 7) + ((u >> 3) & 1));
	const uint8_t* asrc = abits + ((uint64_t)srow * 8 + part * SPT) * 4;
	v16f acc[RBW][CT];
#pragma unroll
	for (int rb = 0; rb < RBW; rb++)
#pragma unroll
		for (int c = 0; c < CT; c++)
#pragma unroll
			for (int i = 0; i < 16; i++) acc[rb][c][i] = 0.f;
	uint32_t a_reg[SPT];
	auto fetch_a = [&](uint32_t ss) {
		const uint8_t* p = asrc + (uint64_t)(gss0 + (ss < n_ss ? ss : n_ss - 1)) * (QN * 32);
		if constexpr (SPT == 8) {
			const v4i v = *reinterpret_cast<const v4i*>(p), w = *reinterpret_cast<const v4i*>(p + 16);
			a_reg[0] = v.x; a_reg[1] = v.y; a_reg[2] = v.z; a_reg[3] = v.w; a_reg[4] = w.x; a_reg[5] = w.y; a_reg[6] = w.z; a_reg[7] = w.w;
		} else if constexpr (SPT == 4) { const v4i v = *reinterpret_cast<const v4i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; a_reg[2] = v.z; a_reg[3] = v.w; }
		else if constexpr (SPT == 2) { const v2i v = *reinterpret_cast<const v2i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; }
		else a_reg[0] = *reinterpret_cast<const uint32_t*>(p);
	};
	auto park = [&](uint32_t buf) {
#pragma unroll
		for (int t = 0; t < SPT; t++) {
			const uint32_t sg = part * SPT + t, w = a_reg[t];
			sA[buf][srow * 8 + (sg ^ ((srow >> 1) & 7))] = v4i{(int)((w << 2) & kM4), (int)(w & kM2), (int)((w >> 2) & kM1), (int)((w >> 1) & kM4)};
		}
	};
	auto fetch_b = [&](int c, uint32_t ss) { return *reinterpret_cast<const v4i*>(brow[c] + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * 1024); };
	auto multiply = [&](uint32_t buf, const v4i (&bq)[CT]) {
		const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			v8i B[CT];
#pragma unroll
			for (int c = 0; c < CT; c++) {
				const uint32_t w = (uint32_t)bq[c][t];
				B[c] = v8i{(int)(w & kM1), (int)(w & kM2), (int)(w & kM4), (int)((w >> 3) & kM1), 0, 0, 0, 0};
			}
#pragma unroll
			for (int rb = 0; rb < RBW; rb++) {
				const v4i a = sA[buf][(32 * (qg * RBW + rb) + r) * 8 + ((2 * t + hh) ^ sw)];
				const v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
#pragma unroll
				for (int c = 0; c < CT; c++) acc[rb][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B[c], acc[rb][c], 4, 4, 0, 0, 0, 0);
			}
		}
	};
	// P2 as in the int8 form: the hot entries of both 128-bin steps of this tile step (by the waves of query group 0: a column tile is
	// held by QS waves)
	auto hotfix = [&](uint32_t ss, const v4i (&bq)[CT]) {
		const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss)]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss) + 2]);
		for (uint32_t e = h0; e < h1; e++) {
			const uint2 en = hot[e];
			const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
			const uint32_t j = (bin >> 5) & 7, wsel = j >> 1;
#pragma unroll
			for (int c = 0; c < CT; c++) {
				const uint32_t w = (uint32_t)(wsel == 0 ? bq[c].x : wsel == 1 ? bq[c].y : wsel == 2 ? bq[c].z : bq[c].w);
				const uint32_t bit = (w >> (16 * (j & 1) + (bin & 15))) & 1u;
				if (valid[c] && (lane >> 5) == ((bin >> 4) & 1) && bit) atomicAdd(out_diff + (uint64_t)ci[c] * QN + (rg >> 16), (int32_t)(rg & 0xffffu));
			}
		}
	};
	v4i bcur[CT], bnext[CT];
#pragma unroll
	for (int c = 0; c < CT; c++) bcur[c] = fetch_b(c, 0);
	fetch_a(0);
	park(0);
	__syncthreads();
	for (uint32_t ss = 0; ss < n_ss; ss++) {
		const uint32_t buf = ss & 1;
#pragma unroll
		for (int c = 0; c < CT; c++) bnext[c] = fetch_b(c, ss + 1);            // (the last one once more past the end: a load nobody uses is cheaper than a branch around it)
		fetch_a(ss + 1);
		multiply(buf, bcur);
		if (hot_ptr && qg == 0) hotfix(ss, bcur);
		park(buf ^ 1);
		__syncthreads();
#pragma unroll
		for (int c = 0; c < CT; c++) bcur[c] = bnext[c];
	}
#pragma unroll
	for (int c = 0; c < CT; c++) {
		if (!valid[c]) continue;
		int32_t* o = out_min + ((uint64_t)ks * m + ci[c]) * QN + 32 * (qg * RBW) + 4 * (lane >> 5);
#pragma unroll
		for (int rb = 0; rb < RBW; rb++)
#pragma unroll
			for (int g = 0; g < 4; g++) *reinterpret_cast<v4i*>(o + 32 * rb + 8 * g) = v4i{(int)acc[rb][c][4 * g], (int)acc[rb][c][4 * g + 1], (int)acc[rb][c][4 * g + 2], (int)acc[rb][c][4 * g + 3]};
	}
}

// The same product with the queries' tile brought into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU) from an image
// k_kb_gather wrote once per block in the tile's own byte order (nibbles, swizzled segments): the 33 vector operations per thread and
// super-step that expanded bits into the tile in every one of the grid's workgroups are gone. A wave copies its QN / NW rows of the
// next super-step (1 KiB per instruction) while the current one is multiplied; vector-memory operations retire in order, so the
// s_waitcnt vmcnt(0) in front of the barrier covers the copy and the candidates' next 16 bytes alike.
__device__ __forceinline__ void lds_dma_1k(const uint8_t* lane_src, uint32_t lds_wave_base) {
	uint32_t keep;      // m0 is the compiler's: saved and restored around the copy
	asm volatile(
	    "s_mov_b32 %0, m0\n\t"
	    "s_mov_b32 m0, %2\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dwordx4 %1, off\n\t"
	    "s_mov_b32 m0, %0"
	    : "=&s"(keep)
	    : "v"(lane_src), "s"(lds_wave_base)
	    : "memory");
}

// The candidates' 16 bytes per lane and super-step come by LDS-DMA too, into a ring of kBDepth KiB per wave, kBDepth - 1 super-steps
// ahead: a register destination can only be one super-step ahead (the wait in front of the barrier would have to let exactly that load
// through, and a register still in flight cannot be handed on), and one super-step -- 0.2-0.5 us -- is less than a trip to HBM.
// Order of issue per super-step: the tile's pieces, THEN the candidates' KiB; vector-memory operations retire in order, so
// s_waitcnt vmcnt(1) in front of the barrier lets only that youngest copy stay in flight.
template <int NRB, int NW>
__global__ void __launch_bounds__(64 * NW, NRB == 8 ? 2 : 4) k_pair_gemm_fp4_dma(const uint8_t* __restrict__ cand_kb, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                                         const uint8_t* __restrict__ anib, uint64_t nbins, uint32_t k_slices, const uint32_t* __restrict__ hot_ptr,
                                                                         const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff, uint32_t n_whole,
                                                                         uint32_t piece_ss) {
	constexpr int QN = 32 * NRB;
	constexpr uint32_t kTile = QN * 128;              // bytes of the queries' tile of a super-step
	constexpr uint32_t kPieces = kTile / NW / 1024;   // KiB a wave copies per super-step
	constexpr uint32_t kBDepth = 4;                   // ring slots of the candidates' bits per wave
	static_assert(kPieces >= 1 && kPieces * NW * 1024 == kTile, "tile copy");
	__shared__ v4i sA[2][QN * 8];
	__shared__ v4i sB[NW][kBDepth][64];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// Workgroups 0 .. n_whole - 1 take a whole tile of 128 candidates each (one slice of the bins: blockIdx.y). With piece_ss > 0 the tiles
	// past n_whole -- what is left when the tiles do not fill the chip's places a whole number of times -- are cut into pieces of piece_ss
	// super-steps, one workgroup each, which ADD their sums into rows the host zeroed: 782 tiles on 768 places ran a second round of 14
	// workgroups on an otherwise idle chip (1.60 ms against 1.06 for 768 tiles); as ~730 short pieces they are one more twentieth of a round.
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices;
	uint32_t tile = blockIdx.x, n_ss = (uint32_t)(per / 256), gss0 = (uint32_t)((uint64_t)ks * per / 256);
	const bool piece = piece_ss && blockIdx.x >= n_whole;
	if (piece) {
		const uint32_t ppt = (n_ss + piece_ss - 1) / piece_ss, p = blockIdx.x - n_whole;
		tile = n_whole + p / ppt;
		gss0 = (p % ppt) * piece_ss;
		n_ss = n_ss - gss0 < piece_ss ? n_ss - gss0 : piece_ss;
	}
	const uint32_t ci = (tile * NW + wave) * 32 + (lane & 31);
	const bool valid = ci < m;
	const uint32_t cc = valid ? ci : m - 1;
	const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
	const uint8_t* brow = cand_kb + (slot >> 5) * msc_kb_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16 + (uint64_t)gss0 * 1024;
	const uint8_t* asrc = anib + (uint64_t)gss0 * kTile + wave * (kPieces * 1024) + lane * 16;
	const uint32_t lds_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sA[0][0] + wave * (kPieces * 1024));
	const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sB[wave][0][0]);
	v16f acc[NRB];
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int i = 0; i < 16; i++) acc[rb][i] = 0.f;
	auto copy_a = [&](uint32_t buf, uint32_t ss) {
		const uint8_t* p = asrc + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * kTile;
#pragma unroll
		for (uint32_t i = 0; i < kPieces; i++) lds_dma_1k(p + i * 1024, lds_a + buf * kTile + i * 1024);
	};
	// a lane's source is its own candidate's 16 bytes: one wave-instruction gathers the 64 of them into one KiB of the ring
	auto copy_b = [&](uint32_t ss) { lds_dma_1k(brow + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * 1024, lds_b + (ss % kBDepth) * 1024); };
	// prologue: the candidates' bits of the first kBDepth - 1 super-steps, the first tile; everything has landed behind the wait
#pragma unroll
	for (uint32_t d = 0; d + 1 < kBDepth; d++) copy_b(d);
	copy_a(0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (uint32_t ss = 0; ss < n_ss; ss++) {
		const uint32_t buf = ss & 1;
		copy_a(buf ^ 1, ss + 1);          // (the buffer the previous super-step read: every wave is past that super-step's barrier)
		copy_b(ss + kBDepth - 1);         // (the slot super-step ss - 1 read)
		const v4i bcur = sB[wave][ss % kBDepth][lane];
		{
			const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const uint32_t w = (uint32_t)bcur[t];
				const v8i B = {(int)(w & kM1), (int)(w & kM2), (int)(w & kM4), (int)((w >> 3) & kM1), 0, 0, 0, 0};
#pragma unroll
				for (int rb = 0; rb < NRB; rb++) {
					const v4i a = sA[buf][(32 * rb + r) * 8 + ((2 * t + hh) ^ sw)];
					const v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
					acc[rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[rb], 4, 4, 0, 0, 0, 0);
				}
			}
		}
		if (hot_ptr) {          // P2 (see k_pair_gemm_fp4)
			const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss)]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss) + 2]);
			for (uint32_t e = h0; e < h1; e++) {
				const uint2 en = hot[e];
				const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
				const uint32_t j = (bin >> 5) & 7, wsel = j >> 1;
				const uint32_t w = (uint32_t)(wsel == 0 ? bcur.x : wsel == 1 ? bcur.y : wsel == 2 ? bcur.z : bcur.w);
				const uint32_t bit = (w >> (16 * (j & 1) + (bin & 15))) & 1u;
				if (valid && (lane >> 5) == ((bin >> 4) & 1) && bit) atomicAdd(out_diff + (uint64_t)ci * QN + (rg >> 16), (int32_t)(rg & 0xffffu));
			}
		}
		asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
		__syncthreads();
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (nothing of this wave may still be writing LDS when the workgroup ends)
	if (!valid) return;
	int32_t* o = out_min + ((uint64_t)ks * m + ci) * QN + 4 * (lane >> 5);
	if (piece) {          // (most pairs of unrelated sequences share no k-mer inside a piece)
#pragma unroll
		for (int rb = 0; rb < NRB; rb++)
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const int v = (int)acc[rb][i];
				if (v) atomicAdd(o + 32 * rb + 8 * (i >> 2) + (i & 3), v);
			}
		return;
	}
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int g = 0; g < 4; g++) *reinterpret_cast<v4i*>(o + 32 * rb + 8 * g) = v4i{(int)acc[rb][4 * g], (int)acc[rb][4 * g + 1], (int)acc[rb][4 * g + 2], (int)acc[rb][4 * g + 3]};
}

// The same product with the WORK, not the output, dealt out ("stream-K"): the pass is tiles x super-steps units of work (a tile = the
// workgroup's 128 candidates, a unit = one 256-bin super-step of it); workgroup w of G takes units [w L, (w + 1) L), L = ceil(units / G),
// i.e. the tail of one tile's bins, whole tiles, the head of another -- every workgroup the same amount whatever the number of candidates,
// and G = what the chip holds at once (4 per CU). A workgroup adds each piece into ONE zeroed int32 array [candidate][QN] with atomics
// (an element gets one or two adds), so there are no slices for the epilogue to add up. With slices the grid was tiles x slices
// workgroups: 782 at 100 000 candidates and one slice (three quarters of the chip's 1 024 places), and a small shard needed 8 - 64 slices
// to fill the chip at all.
template <int NRB, int NW>
__global__ void __launch_bounds__(64 * NW, NRB == 8 ? 2 : 4) k_pair_gemm_fp4_sk(const uint8_t* __restrict__ cand_kb, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                                 const uint8_t* __restrict__ abits, uint64_t nbins, uint32_t per_wg, const uint32_t* __restrict__ hot_ptr,
                                                                 const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff) {
	constexpr int QN = 32 * NRB;
	constexpr int NT = 64 * NW;
	constexpr int SPT = QN * 8 / NT;
	constexpr int TPR = 8 / SPT;
	static_assert(SPT >= 1 && SPT <= 8 && TPR * SPT == 8, "tile staging");
	__shared__ v4i sA[2][QN * 8];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t n_ss = (uint32_t)(nbins / 256);
	const uint32_t tiles = (m + 32 * NW - 1) / (32 * NW);
	const uint64_t total = (uint64_t)tiles * n_ss;
	uint64_t pos = (uint64_t)blockIdx.x * per_wg;
	const uint64_t pos_end = pos + per_wg < total ? pos + per_wg : total;
	const uint32_t u = tid / TPR, part = tid % TPR;
	const uint32_t srow = (u & ~15u) | (2 * (u & 7) + ((u >> 3) & 1));
	const uint8_t* asrc = abits + ((uint64_t)srow * 8 + part * SPT) * 4;
	uint32_t a_reg[SPT];
	auto fetch_a = [&](uint32_t ss) {
		const uint8_t* p = asrc + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * (QN * 32);
		if constexpr (SPT == 8) {
			const v4i v = *reinterpret_cast<const v4i*>(p), w = *reinterpret_cast<const v4i*>(p + 16);
			a_reg[0] = v.x; a_reg[1] = v.y; a_reg[2] = v.z; a_reg[3] = v.w; a_reg[4] = w.x; a_reg[5] = w.y; a_reg[6] = w.z; a_reg[7] = w.w;
		} else if constexpr (SPT == 4) { const v4i v = *reinterpret_cast<const v4i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; a_reg[2] = v.z; a_reg[3] = v.w; }
		else if constexpr (SPT == 2) { const v2i v = *reinterpret_cast<const v2i*>(p); a_reg[0] = v.x; a_reg[1] = v.y; }
		else a_reg[0] = *reinterpret_cast<const uint32_t*>(p);
	};
	auto park = [&](uint32_t buf) {
#pragma unroll
		for (int t = 0; t < SPT; t++) {
			const uint32_t sg = part * SPT + t, w = a_reg[t];
			sA[buf][srow * 8 + (sg ^ ((srow >> 1) & 7))] = v4i{(int)((w << 2) & kM4), (int)(w & kM2), (int)((w >> 2) & kM1), (int)((w >> 1) & kM4)};
		}
	};
	while (pos < pos_end) {
		const uint32_t tile = (uint32_t)(pos / n_ss), ss0 = (uint32_t)(pos % n_ss);
		const uint32_t ss1 = (uint32_t)((uint64_t)(n_ss - ss0) < pos_end - pos ? n_ss : ss0 + (pos_end - pos));          // this piece: super-steps [ss0, ss1) of the tile
		const uint32_t ci = (tile * NW + wave) * 32 + (lane & 31);
		const bool valid = ci < m;
		const uint32_t cc = valid ? ci : m - 1;
		const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
		const uint8_t* brow = cand_kb + (slot >> 5) * msc_kb_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16;
		auto fetch_b = [&](uint32_t ss) { return *reinterpret_cast<const v4i*>(brow + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * 1024); };
		v16f acc[NRB];
#pragma unroll
		for (int rb = 0; rb < NRB; rb++)
#pragma unroll
			for (int i = 0; i < 16; i++) acc[rb][i] = 0.f;
		v4i bcur = fetch_b(ss0);
		fetch_a(ss0);
		__syncthreads();          // (the previous piece's last reads of sA are through)
		park(0);
		__syncthreads();
		for (uint32_t ss = ss0; ss < ss1; ss++) {
			const uint32_t buf = (ss - ss0) & 1;
			const v4i bnext = fetch_b(ss + 1);
			fetch_a(ss + 1);
			{
				const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
				for (int t = 0; t < 4; t++) {
					const uint32_t w = (uint32_t)bcur[t];
					const v8i B = {(int)(w & kM1), (int)(w & kM2), (int)(w & kM4), (int)((w >> 3) & kM1), 0, 0, 0, 0};
#pragma unroll
					for (int rb = 0; rb < NRB; rb++) {
						const v4i a = sA[buf][(32 * rb + r) * 8 + ((2 * t + hh) ^ sw)];
						const v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
						acc[rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[rb], 4, 4, 0, 0, 0, 0);
					}
				}
			}
			if (hot_ptr) {          // P2 (see k_pair_gemm_fp4): the hot entries of both 128-bin steps of this super-step
				const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * ss]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * ss + 2]);
				for (uint32_t e = h0; e < h1; e++) {
					const uint2 en = hot[e];
					const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
					const uint32_t j = (bin >> 5) & 7, wsel = j >> 1;
					const uint32_t w = (uint32_t)(wsel == 0 ? bcur.x : wsel == 1 ? bcur.y : wsel == 2 ? bcur.z : bcur.w);
					const uint32_t bit = (w >> (16 * (j & 1) + (bin & 15))) & 1u;
					if (valid && (lane >> 5) == ((bin >> 4) & 1) && bit) atomicAdd(out_diff + (uint64_t)ci * QN + (rg >> 16), (int32_t)(rg & 0xffffu));
				}
			}
			park(buf ^ 1);
			__syncthreads();
			bcur = bnext;
		}
		if (valid) {
			int32_t* o = out_min + (uint64_t)ci * QN + 4 * (lane >> 5);
#pragma unroll
			for (int rb = 0; rb < NRB; rb++)
#pragma unroll
				for (int i = 0; i < 16; i++) {
					const int v = (int)acc[rb][i];
					if (v) atomicAdd(o + 32 * rb + 8 * (i >> 2) + (i & 3), v);          // (most pairs of unrelated sequences share no k-mer in a piece)
				}
		}
		pos += ss1 - ss0;
	}
}

}  // namespace

// MSC_GEMM_I8 keeps the int8 form (k_pair_gemm_bits) for A/B runs; the FP4 form is the default
static bool pair_gemm_fp4() {
	static const bool i8 = getenv("MSC_GEMM_I8") != nullptr;
	return !i8;
}
// MSC_GEMM_NO_DMA: the queries' tile expanded from bits by every workgroup (k_pair_gemm_fp4) instead of copied by LDS-DMA (k_pair_gemm_fp4_dma)
static bool pair_gemm_dma() {
	static const bool off = getenv("MSC_GEMM_NO_DMA") != nullptr || getenv("MSC_GEMM_QS2") != nullptr || getenv("MSC_GEMM_WAVES") != nullptr || getenv("MSC_GEMM_STREAMK") != nullptr;
	return pair_gemm_fp4() && !off;
}
// MSC_GEMM_STREAMK: the work-dealing grid (k_pair_gemm_fp4_sk) instead of the sliced one. Measured SLOWER and therefore off: 4.85 against
// 5.11 G pairs/s at 100 000 candidates (the product alone 1.38 against 1.49 ms, but a grid that fills every place of the chip leaves the
// tail stream's kernels nowhere to run: 2.04 ms beside them), 3.41 against 3.74 at 12 500; 2 / 3 / 6 / 8 workgroups per CU
// (MSC_GEMM_SK_PER_CU) no better. Kept for A/B runs.
static bool pair_gemm_streamk() {
	static const bool on = getenv("MSC_GEMM_STREAMK") != nullptr && getenv("MSC_GEMM_QS2") == nullptr && getenv("MSC_GEMM_WAVES") == nullptr;
	return pair_gemm_fp4() && on;
}
const char* msc_pair_gemm_kernel_name() { return pair_gemm_streamk() ? "k_pair_gemm_fp4_sk" : pair_gemm_dma() ? "k_pair_gemm_fp4_dma" : pair_gemm_fp4() ? "k_pair_gemm_fp4" : "k_pair_gemm_bits"; }

uint64_t msc_kb_bytes(const MscLayout& L, uint64_t capacity) { return (capacity + 31) / 32 * msc_kb_block_bytes(L.padded_bins); }

// flags (device, two int32 zeroed by the caller): [0] a zero count was met, [1] the longest list of large bins
hipError_t msc_launch_kb_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, uint8_t* kb, uint64_t first_slot, uint64_t n_slots, void* mb,
                               uint32_t* mb_n, uint32_t pitch, int32_t* flags) {
	if (n_slots == 0) return hipSuccess;
	if (L.padded_bins % 256) return hipErrorInvalidValue;
	hipError_t e = hipMemsetAsync(mb_n + first_slot, 0, n_slots * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	const uint64_t threads = n_slots * L.padded_bins / 16;
	const dim3 grid((unsigned)((threads + 255) / 256));
	if (dtype == 8) k_kb_build<uint8_t><<<grid, dim3(256), 0, st>>>((const uint8_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else if (dtype == 16) k_kb_build<uint16_t><<<grid, dim3(256), 0, st>>>((const uint16_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else if (dtype == 32) k_kb_build<uint32_t><<<grid, dim3(256), 0, st>>>((const uint32_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

// rows of queries one pass serves for a block of n_q: 32, 64, 128 or 256
uint32_t msc_pair_gemm_rows(uint32_t n_q) { return n_q <= 32 ? 32 : n_q <= 64 ? 64 : n_q <= 128 ? 128 : 256; }

// 128 candidates per workgroup (4 waves). MSC_GEMM_WAVES=8 selects the 8-wave form (256 candidates: the queries' tile is expanded into LDS
// once for twice the products) -- measured 2 % SLOWER over 100 000 candidates (3.33 against 3.40 G pairs/s: its barriers hold eight waves
// instead of four), kept as a variant for A/B runs.
static bool pair_gemm_wide(uint32_t qn, uint32_t m) {
	static const int nw_env = [] { const char* e = getenv("MSC_GEMM_WAVES"); return e ? atoi(e) : 0; }();
	return qn == 128 && nw_env == 8 && m >= 16384;
}

uint32_t msc_pair_gemm_slices(uint64_t nbins, uint32_t m, uint32_t qn, int num_cus) {
	// As few slices as give every CU about three workgroups (12 waves of its 16): the first rule here asked for six ROUNDS of the chip
	// (24 workgroups per CU) and paid for them twice -- every slice is another int32 per pair written by the product and read and added
	// by the epilogue, and a workgroup's set-up is spread over fewer steps. Measured (G pairs/s, cfg2's sequences, FP4 form):
	//   100 000 candidates (782 workgroups per slice): 1 / 2 / 4 / 8 slices -> 5.10 / 5.08 / 4.97 / 4.76
	//    50 000: 2 / 4 -> 4.88 / 4.77        25 000: 2 / 4 / 8 -> 3.55 / 4.32 / 4.16        12 500: 4 / 8 / 16 / 32 / 64 -> 3.29 / 3.72 / 3.62 / 3.12 / 2.40
	// -- the best is always the count that brings the grid to ~780 workgroups. A slice is an even number of 128-bin steps.
	// MSC_GEMM_MAX_SLICES / MSC_GEMM_SLICES bound the count from above / below for A/B runs.
	if (pair_gemm_streamk()) return 1;          // (the work is dealt out by units, the output is one array)
	const uint32_t per_wg = pair_gemm_wide(qn, m) ? 256 : 128;
	uint32_t s = 1;
	static const uint32_t s_max = [] { const char* e = getenv("MSC_GEMM_MAX_SLICES"); return (uint32_t)(e && atoi(e) > 0 ? atoi(e) : 64); }();
	auto can_split = [&] { return s < s_max && nbins / (2 * s) >= 2 * kStep && nbins % (2 * s * 2 * kStep) == 0; };
	while (can_split() && (uint64_t)((m + per_wg - 1) / per_wg) * s < (uint64_t)num_cus * (per_wg == 256 ? 2 : 3)) s *= 2;
	static const uint32_t s_min = [] { const char* e = getenv("MSC_GEMM_SLICES"); return (uint32_t)(e ? std::max(1, atoi(e)) : 1); }();
	while (can_split() && s < s_min) s *= 2;
	return s;
}

// bytes of the queries' side of a block: the bit image (16 bytes per row and step) and the transposed counts (a byte per bin and row)
uint64_t msc_pair_gemm_abits_bytes(uint64_t nbins, uint32_t qn) { return nbins / 8 * qn; }
uint64_t msc_pair_gemm_qt_bytes(uint64_t nbins, uint32_t qn) { return nbins * 2 * (qn < 128 ? 16 : qn / 8); }
// ... and the nibble image of the queries' tiles for the LDS-DMA form of the product (0: that form is off)
uint64_t msc_pair_gemm_anib_bytes(uint64_t nbins, uint32_t qn) { return pair_gemm_dma() ? nbins / 2 * qn : 0; }

// The queries' side: abits and qT of rows q_slots[0 .. n_q) of the mirror q_kb; when n_hot > 0 also the hot list (hot: n_hot entries;
// hot_ptr, hot_cursor, hot_cnt: nbins / 128 + 1 words each) from the queries' lists of large bins.
hipError_t msc_launch_pair_gemm_queries(hipStream_t st, uint64_t nbins, const uint8_t* q_kb, const void* q_mb, const uint32_t* q_mb_n, uint32_t q_pitch,
                                        const uint32_t* q_slots_dev, uint32_t n_q, uint32_t qn, uint8_t* abits, uint8_t* qT, uint64_t n_hot, void* hot, uint32_t* hot_ptr,
                                        uint32_t* hot_cursor, uint32_t* hot_cnt, uint8_t* anib) {
	if (n_q == 0 || n_q > qn || nbins % 256) return hipErrorInvalidValue;
	const uint32_t nsteps = (uint32_t)(nbins / kStep);
	const int fp4 = pair_gemm_fp4() ? 1 : 0;
	if (qn == 32) k_kb_gather<32><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, abits, qT, fp4, pair_gemm_dma() ? anib : nullptr);
	else if (qn == 64) k_kb_gather<64><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, abits, qT, fp4, pair_gemm_dma() ? anib : nullptr);
	else if (qn == 128) k_kb_gather<128><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, abits, qT, fp4, pair_gemm_dma() ? anib : nullptr);
	else if (qn == 256) k_kb_gather<256><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, abits, qT, fp4, pair_gemm_dma() ? anib : nullptr);
	else return hipErrorInvalidValue;
	hipError_t e = hipGetLastError();
	if (e != hipSuccess || n_hot == 0) return e;
	if ((e = hipMemsetAsync(hot_cnt, 0, (nsteps + 1) * sizeof(uint32_t), st)) != hipSuccess) return e;
	k_hot_count<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, hot_cnt);
	k_hot_scan<<<dim3(1), dim3(1024), 0, st>>>(hot_cnt, nsteps, hot_ptr, hot_cursor);
	k_hot_fill<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, qn, hot_cursor, (uint2*)hot, qT);
	return hipGetLastError();
}

// P1 [k_slices][m][qn] and, with a hot list, P2 [m][qn] (zeroed here) of the block's queries against m candidates (slot list, or slots
// first .. first + m - 1) of the mirror cand_kb
hipError_t msc_launch_pair_gemm(hipStream_t st, uint64_t nbins, const uint8_t* cand_kb, const uint32_t* cand_slots, uint64_t first, uint32_t m, const uint8_t* abits,
                                uint32_t qn, uint32_t k_slices, const uint32_t* hot_ptr, const void* hot, int32_t* out_min, int32_t* out_diff, const uint8_t* anib) {
	if (m == 0) return hipSuccess;
	if (k_slices == 0 || nbins % ((uint64_t)k_slices * 2 * kStep)) return hipErrorInvalidValue;
	if (hot_ptr) {
		const hipError_t e = hipMemsetAsync(out_diff, 0, (size_t)m * qn * sizeof(int32_t), st);
		if (e != hipSuccess) return e;
	}
	if (pair_gemm_dma() && anib) {
		static const unsigned dma_pad = [] { const char* e = getenv("MSC_GEMM_LDS_PAD"); return (unsigned)(e ? atoi(e) * 1024 : 0); }();
		// the tiles that do not fill the chip's places a whole number of times go as short pieces (see the kernel); one slice only
		static const bool no_pieces = getenv("MSC_GEMM_NO_PIECES") != nullptr;
		static const int num_cus = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256; return p.multiProcessorCount; }();
		const uint32_t tiles = (m + 127) / 128, places = (uint32_t)num_cus * (qn == 256 ? 2 : 3), n_ss_all = (uint32_t)(nbins / 256);
		uint32_t n_whole = tiles, piece_ss = 0, grid_x = tiles;
		if (k_slices == 1 && !no_pieces && tiles > places && tiles % places && (tiles % places) * 4 <= places) {          // (a leftover of more than a quarter round is better off as a round of whole tiles: 1.87 against 1.95 ms at 150 000 candidates)
			n_whole = tiles / places * places;
			const uint32_t left = tiles - n_whole;
			piece_ss = std::max<uint32_t>(16, (uint32_t)(((uint64_t)left * n_ss_all + places - 1) / places));
			if (piece_ss >= n_ss_all) { n_whole = tiles; piece_ss = 0; }
			else {
				grid_x = n_whole + left * ((n_ss_all + piece_ss - 1) / piece_ss);
				const hipError_t e = hipMemsetAsync(out_min + (size_t)n_whole * 128 * qn, 0, (size_t)(m - n_whole * 128) * qn * sizeof(int32_t), st);
				if (e != hipSuccess) return e;
			}
		}
		const dim3 grid(grid_x, k_slices);
#define MSC_DMA_GO(NRB) k_pair_gemm_fp4_dma<NRB, 4><<<grid, dim3(256), dma_pad, st>>>(cand_kb, cand_slots, first, m, anib, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff, n_whole, piece_ss)
		if (qn == 32) MSC_DMA_GO(1);
		else if (qn == 64) MSC_DMA_GO(2);
		else if (qn == 128) MSC_DMA_GO(4);
		else if (qn == 256) MSC_DMA_GO(8);
		else return hipErrorInvalidValue;
#undef MSC_DMA_GO
		return hipGetLastError();
	}
	if (pair_gemm_streamk()) {
		if (k_slices != 1) return hipErrorInvalidValue;
		hipError_t e = hipMemsetAsync(out_min, 0, (size_t)m * qn * sizeof(int32_t), st);
		if (e != hipSuccess) return e;
		const uint64_t total = (uint64_t)((m + 127) / 128) * (nbins / 256);
		static const int num_cus = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256; return p.multiProcessorCount; }();
		static const int per_cu = [] { const char* e = getenv("MSC_GEMM_SK_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
		const uint64_t places = (uint64_t)num_cus * per_cu;          // four workgroups of four waves per CU at most (108 registers, 32 KiB of LDS each)
		const uint32_t per_wg = (uint32_t)((total + places - 1) / places);
		const dim3 grid((unsigned)((total + per_wg - 1) / per_wg));
#define MSC_SK_GO(NRB) k_pair_gemm_fp4_sk<NRB, 4><<<grid, dim3(256), 0, st>>>(cand_kb, cand_slots, first, m, abits, nbins, per_wg, hot_ptr, (const uint2*)hot, out_min, out_diff)
		if (qn == 32) MSC_SK_GO(1);
		else if (qn == 64) MSC_SK_GO(2);
		else if (qn == 128) MSC_SK_GO(4);
		else if (qn == 256) MSC_SK_GO(8);
		else return hipErrorInvalidValue;
#undef MSC_SK_GO
		return hipGetLastError();
	}
	const bool wide = pair_gemm_wide(qn, m);
	// MSC_GEMM_LDS_PAD=KiB: unused LDS added to every workgroup of the product, i.e. fewer of them per CU -- room for the tail stream's kernels
	static const bool qs2 = getenv("MSC_GEMM_QS2") != nullptr;          // (A/B: the 64-candidate-per-wave deal of the workgroup's tile)
	static const unsigned lds_pad = [] { const char* e = getenv("MSC_GEMM_LDS_PAD"); return (unsigned)(e ? atoi(e) * 1024 : 0); }();
	const dim3 grid((m + (wide ? 255 : 127)) / (wide ? 256 : 128), k_slices);
#define MSC_PG_GO(NRB, NW)                                                                                                                                             \
	do {                                                                                                                                                           \
		if (pair_gemm_fp4() && qs2 && NRB % 2 == 0) k_pair_gemm_fp4<NRB, NW, (NRB % 2 == 0 ? 2 : 1)><<<grid, dim3(64 * NW), lds_pad, st>>>(cand_kb, cand_slots, first, m, abits, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff); \
		else if (pair_gemm_fp4()) k_pair_gemm_fp4<NRB, NW, 1><<<grid, dim3(64 * NW), lds_pad, st>>>(cand_kb, cand_slots, first, m, abits, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff); \
		else k_pair_gemm_bits<NRB, NW><<<grid, dim3(64 * NW), 0, st>>>(cand_kb, cand_slots, first, m, abits, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff);             \
	} while (0)
	if (qn == 32) MSC_PG_GO(1, 4);
	else if (qn == 64) MSC_PG_GO(2, 4);
	else if (qn == 128 && wide) MSC_PG_GO(4, 8);
	else if (qn == 128) MSC_PG_GO(4, 4);
	else if (qn == 256) MSC_PG_GO(8, 4);
	else return hipErrorInvalidValue;
#undef MSC_PG_GO
	return hipGetLastError();
}
