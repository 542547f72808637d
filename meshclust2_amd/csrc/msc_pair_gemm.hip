// msc_pair_gemm.hip -- the Q x M pass on the matrix cores (gfx950 MFMA), r04 form: ONE int8 matrix product per tile of bins, exact
// for ANY counts.
//
// The pass needs three integer reductions per (query q, candidate c) (pair_features.hip): sum |q_i - c_i| (manhattan, intersection,
// kulczynski2: predict/Feature.cpp:859-871,764-777,682-695), sum q_i c_i (euclidean, normalized_vectors, pearson, simratio:
// :1113-1124,1171-1184,795-811,829-841) and the earth mover's distance (:1505-1518; msc_emd_ranks.hip). With excess counts
// e = count - 1 (every bin starts at the pseudocount 1, nonltr/KmerHashTable.cpp:69-72):
//     sum |q_i - c_i| = sum e_q + sum e_c - 2 sum min(e_q, e_c)          sum q_i c_i = 4^k + sum e_q + sum e_c + sum e_q e_c
// An L-base sequence touches at most L of the 4^k bins, almost all of them ONCE: a 1 kb sequence at k = 9 has ~2 bins with e >= 2.
// Where the candidate's excess is 0 or 1, min(e_q, e_c) = [e_q >= 1] e_c -- bilinear. So with the clamped bytes x = min(e, 127) and the
// queries' flag bytes f = [e >= 1]:
//     P1(q, c) = sum_i f_q(i) x_c(i)                  one v_mfma_i32_32x32x32_i8 per 32 queries x 32 candidates x 32 bins, nothing else
//     sum min(e_q, e_c) = P1 + sum over the candidate's bins with e_c >= 2 of [ min(e_q, e_c) - f_q min(e_c, 127) ]
//     sum e_q e_c       = P1 + P2 + sum over bins with e_q > 127 or e_c > 127 of [ e_q e_c - x_q x_c ],   P2 = sum_i (x_q(i) - f_q(i)) x_c(i)
// The corrections run over SHORT LISTS, not over bins: per slot the (bin, e) pairs with e >= 2 ("large bins": ~2 per 1 kb sequence at
// k = 9, one more per repeat unit), kept beside the mirror. The candidate's list is walked by the epilogue (k_pair_epilogue_x8 in
// pair_features.hip: one coalesced read of the queries' bytes at that bin per entry); the queries' lists become a per-step "hot list"
// this kernel consults while the candidates' bytes of that step sit in its registers (P2: one atomic add per (entry, candidate that
// holds the k-mer) -- 0.4 % of the candidates at k = 9). Exact in integers for any counts the narrow range admits (<= 8191): the r03
// form (thermometer levels) ran only while the LARGEST count of both whole sets was <= 16 -- one homopolymer run among 100 000
// sequences sent every pair to the digest kernel at a fifth of the rate.
//
//   x8 mirror       msc_x8.h: a byte per bin, slots blocked by 32; + the lists of large bins (mb, pitch entries per slot, unordered)
//   k_x8_gather     the queries' side of a block of <= QN queries: the flag bytes as the LDS image the GEMM stages (16-byte segments
//                   XOR-swizzled so that a 32-row A operand read is conflict-free), and the clamped bytes transposed [bin][query] for
//                   the epilogue's lookups
//   k_hot_*         the queries' large bins bucketed by 128-bin step: (bin, query row, x - 1)
//   k_pair_gemm_x8  workgroup = 128 candidates x QN queries x one slice of the bins; wave = 32 candidates x QN queries: QN / 32
//                   accumulators of 32 x 32. Candidate bytes go from HBM straight into the B operand registers, one step ahead; the
//                   queries' tile of a step (QN x 128 bytes) is staged once per workgroup in LDS. Per 32 x 32 x 32 tile: one
//                   ds_read_b128 + one MFMA. Roofline: HBM -- a candidate byte is read once per QN queries.
//   output          int32 P1 [slice][candidate][QN] (plain stores, the epilogue adds the slices), int32 P2 [candidate][QN] (atomics)
#include "msc_internal.h"
#include "msc_wave.h"
#include "msc_x8.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr uint32_t kStep = 128;          // bins per step

// ------------------------------------------------------------------------------------------------ the mirror and its lists
// 16 bins per thread. flags[0] |= 1 when a zero count is met (the identities above need every count >= 1: KmerHashTable's initial
// value, and a mean of such histograms too); flags[1] = the longest list seen (the host re-lays the lists out when it passes the pitch).
template <typename T>
__global__ void __launch_bounds__(256) k_x8_build(const T* __restrict__ bins, uint64_t nbins, uint8_t* __restrict__ x8, uint64_t first_slot, uint64_t n_slots,
                                                  uint2* __restrict__ mb, uint32_t* __restrict__ mb_n, uint32_t* __restrict__ mb_big, uint32_t pitch, int32_t* __restrict__ flags) {
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (i >= n_slots * nbins) return;
	const uint64_t slot = first_slot + i / nbins, at = i % nbins;
	constexpr int NV = sizeof(T);          // 16-byte loads for 16 bins
	uint32_t raw[4 * NV];
	const uint4* src = reinterpret_cast<const uint4*>(bins + slot * nbins + at);
#pragma unroll
	for (int v = 0; v < NV; v++) { const uint4 q = src[v]; raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w; }
	uint32_t w[4] = {0, 0, 0, 0};
	bool zero = false;
#pragma unroll
	for (int j = 0; j < 16; j++) {
		uint32_t c;
		if constexpr (sizeof(T) == 1) c = (raw[j >> 2] >> (8 * (j & 3))) & 0xffu;
		else if constexpr (sizeof(T) == 2) c = (raw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
		else c = raw[j];
		if (c == 0) { zero = true; continue; }
		const uint32_t e = c - 1;
		w[j >> 2] |= (e > MSC_X8_CAP ? MSC_X8_CAP : e) << (8 * (j & 3));
		if (e >= 2) {
			const uint32_t pos = atomicAdd(&mb_n[slot], 1u);
			if (pos < pitch) mb[slot * pitch + pos] = make_uint2((uint32_t)at + j, e);
			atomicMax(&flags[1], (int32_t)(pos + 1));
			if (e >= MSC_X8_CAP) atomicAdd(&mb_big[slot], 1u);
		}
	}
	*reinterpret_cast<uint4*>(x8 + msc_x8_offset(slot, at, nbins)) = make_uint4(w[0], w[1], w[2], w[3]);
	if (zero) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------------ the queries' side of a block
// One workgroup per 128-bin step. fimg[step][row][seg ^ ((row >> 1) & 7)] (16-byte segments; rows past n_q are zero) is the image
// k_pair_gemm_x8 copies into LDS as it stands; qT[bin][row] the clamped bytes for the epilogue.
template <int QN>
__global__ void __launch_bounds__(256) k_x8_gather(const uint8_t* __restrict__ x8, const uint32_t* __restrict__ q_slots, uint32_t n_q, uint64_t nbins,
                                                   uint8_t* __restrict__ fimg, uint8_t* __restrict__ qT) {
	__shared__ v4i tile[QN * 8];          // [row][segment]: the clamped bytes of this step
	const uint32_t step = blockIdx.x;
	for (uint32_t it = threadIdx.x; it < QN * 8; it += 256) {
		const uint32_t row = it >> 3, s = it & 7;
		v4i v = {0, 0, 0, 0};
		if (row < n_q) v = *reinterpret_cast<const v4i*>(x8 + msc_x8_offset(q_slots[row], (uint64_t)step * kStep + 16 * s, nbins));
		tile[it] = v;
		v4i f;          // [x >= 1] per byte: x <= 127, so x + 127 carries into bit 7 of its own byte only
#pragma unroll
		for (int c = 0; c < 4; c++) f[c] = (int)((((uint32_t)v[c] + 0x7f7f7f7fu) >> 7) & 0x01010101u);
		*reinterpret_cast<v4i*>(fimg + ((uint64_t)step * QN + row) * kStep + ((s ^ ((row >> 1) & 7)) * 16)) = f;
	}
	__syncthreads();
	const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
	for (uint32_t it = threadIdx.x; it < kStep * (QN / 16); it += 256) {
		const uint32_t p = it / (QN / 16), r0 = (it % (QN / 16)) * 16;
		uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < 16; j++) w[j >> 2] |= (uint32_t)tb[(r0 + j) * kStep + p] << (8 * (j & 3));
		*reinterpret_cast<uint4*>(qT + ((uint64_t)step * kStep + p) * QN + r0) = make_uint4(w[0], w[1], w[2], w[3]);
	}
}

// The queries' large bins, bucketed by step: count, exclusive scan, fill. An entry = (bin, row << 8 | x - 1), x = min(e, 127) >= 2.
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
                                                  uint32_t n_q, uint32_t* __restrict__ cursor, uint2* __restrict__ hot) {
	const uint32_t row = blockIdx.x;
	if (row >= n_q) return;
	const uint32_t slot = q_slots[row];
	const uint32_t n = mb_n[slot] < pitch ? mb_n[slot] : pitch;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint2 en = mb[(uint64_t)slot * pitch + i];
		const uint32_t x = en.y > MSC_X8_CAP ? MSC_X8_CAP : en.y;
		hot[atomicAdd(&cursor[en.x / kStep], 1u)] = make_uint2(en.x, (row << 8) | (x - 1));
	}
}

// ------------------------------------------------------------------------------------------------ the product
// NRB = QN / 32 row blocks of queries. Registers: 16 NRB accumulators + two operand sets of 4 x 4 + NRB x 4 of staging: 3 waves per
// SIMD up to QN = 128, 2 at 256.
template <int NRB>
__global__ void __launch_bounds__(256, NRB == 8 ? 2 : 3) k_pair_gemm_x8(const uint8_t* __restrict__ cand8, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                      const uint8_t* __restrict__ fimg, uint64_t nbins, uint32_t k_slices, const uint32_t* __restrict__ hot_ptr,
                                                      const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff) {
	constexpr int QN = 32 * NRB;
	__shared__ v4i sA[2][QN * 8];          // [buffer][row][16-byte segment ^ ((row >> 1) & 7)]: QN x 128 bytes each
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices, k0 = (uint64_t)ks * per;
	const uint32_t steps = (uint32_t)(per / kStep), gstep0 = (uint32_t)(k0 / kStep);
	const uint32_t ci = (blockIdx.x * 4 + wave) * 32 + (lane & 31);
	const bool valid = ci < m;
	const uint32_t cc = valid ? ci : m - 1;
	const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
	// lane l: candidate l % 32 of the block, bytes 16 (l / 32) .. + 15 of every 32-bin chunk: consecutive slots of one block make the
	// wave's load one contiguous KiB
	const uint8_t* brow = cand8 + (slot >> 5) * msc_x8_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16;
	const uint8_t* asrc = fimg + (uint64_t)tid * 16;
	v16i acc[NRB];
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int i = 0; i < 16; i++) acc[rb][i] = 0;
	v4i a_reg[NRB], b0[4], b1[4];
	// operands of step i (the last step once more past the end: a load nobody uses is cheaper than a branch around it)
	auto fetch = [&](uint32_t i, v4i (&b)[4]) {
		const uint32_t j = i < steps ? i : steps - 1;
		const uint64_t k = k0 + (uint64_t)j * kStep;
#pragma unroll
		for (int t = 0; t < NRB; t++) a_reg[t] = *reinterpret_cast<const v4i*>(asrc + (uint64_t)(gstep0 + j) * (QN * kStep) + 4096 * t);
#pragma unroll
		for (int kc = 0; kc < 4; kc++) b[kc] = *reinterpret_cast<const v4i*>(brow + ((k >> 5) + kc) * 1024);
	};
	auto park = [&](uint32_t buf) {
#pragma unroll
		for (int t = 0; t < NRB; t++) sA[buf][tid + 256 * t] = a_reg[t];
	};
	auto multiply = [&](uint32_t buf, const v4i (&b)[4]) {
		const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
		for (int kc = 0; kc < 4; kc++)
#pragma unroll
			for (int rb = 0; rb < NRB; rb++) {
				// A operand: lane l = query 32 rb + l % 32, the bins of half l / 32 of this 32-bin chunk
				const v4i A = sA[buf][(32 * rb + r) * 8 + ((2 * kc + hh) ^ sw)];
				acc[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, b[kc], acc[rb], 0, 0, 0);
			}
	};
	// P2: the queries' large bins that fall into this step, while the candidates' bytes of the step are in registers. Wave-uniform
	// entries; the lane that holds (its candidate, that bin) adds (x_q - 1) x_c for its pair unless x_c is 0 (99.6 % of them at k = 9).
	auto hotfix = [&](uint32_t i, const v4i (&b)[4]) {
		const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[gstep0 + i]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[gstep0 + i + 1]);
		for (uint32_t e = h0; e < h1; e++) {
			const uint2 en = hot[e];
			const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
			const uint32_t kc = (bin >> 5) & 3, byte = bin & 15;
			const v4i bj = kc == 0 ? b[0] : kc == 1 ? b[1] : kc == 2 ? b[2] : b[3];
			const uint32_t wsel = byte >> 2;
			const uint32_t w = (uint32_t)(wsel == 0 ? bj.x : wsel == 1 ? bj.y : wsel == 2 ? bj.z : bj.w);
			const uint32_t val = (w >> (8 * (byte & 3))) & 0xffu;
			if (valid && (lane >> 5) == ((bin >> 4) & 1) && val) atomicAdd(out_diff + (uint64_t)ci * QN + (rg >> 8), (int32_t)((rg & 0xffu) * val));
		}
	};
	fetch(0, b0);
	park(0);
	__syncthreads();
	for (uint32_t i = 0; i < steps; i += 2) {          // two steps per turn (the host makes `steps` even): the operand registers and LDS halves swap roles by name
		fetch(i + 1, b1);
		multiply(0, b0);
		if (hot_ptr) hotfix(i, b0);
		park(1);
		__syncthreads();
		fetch(i + 2, b0);
		multiply(1, b1);
		if (hot_ptr) hotfix(i + 1, b1);
		park(0);
		__syncthreads();
	}
	// D: lane l holds column l % 32 (its candidate); register 4 g + j = row 8 g + 4 (l / 32) + j of each 32-query block
	if (!valid) return;
	int32_t* o = out_min + ((uint64_t)ks * m + ci) * QN + 4 * (lane >> 5);
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int g = 0; g < 4; g++) *reinterpret_cast<v4i*>(o + 32 * rb + 8 * g) = v4i{acc[rb][4 * g], acc[rb][4 * g + 1], acc[rb][4 * g + 2], acc[rb][4 * g + 3]};
}

}  // namespace

uint64_t msc_x8_bytes(const MscLayout& L, uint64_t capacity) { return (capacity + 31) / 32 * msc_x8_block_bytes(L.padded_bins); }

// flags (device, two int32 zeroed by the caller): [0] a zero count was met, [1] the longest list of large bins
hipError_t msc_launch_x8_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, uint8_t* x8, uint64_t first_slot, uint64_t n_slots, void* mb,
                               uint32_t* mb_n, uint32_t* mb_big, uint32_t pitch, int32_t* flags) {
	if (n_slots == 0) return hipSuccess;
	if (L.padded_bins % 16) return hipErrorInvalidValue;
	hipError_t e = hipMemsetAsync(mb_n + first_slot, 0, n_slots * sizeof(uint32_t), st);
	if (e == hipSuccess) e = hipMemsetAsync(mb_big + first_slot, 0, n_slots * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	const uint64_t threads = n_slots * L.padded_bins / 16;
	const dim3 grid((unsigned)((threads + 255) / 256));
	if (dtype == 8) k_x8_build<uint8_t><<<grid, dim3(256), 0, st>>>((const uint8_t*)bins, L.padded_bins, x8, first_slot, n_slots, (uint2*)mb, mb_n, mb_big, pitch, flags);
	else if (dtype == 16) k_x8_build<uint16_t><<<grid, dim3(256), 0, st>>>((const uint16_t*)bins, L.padded_bins, x8, first_slot, n_slots, (uint2*)mb, mb_n, mb_big, pitch, flags);
	else if (dtype == 32) k_x8_build<uint32_t><<<grid, dim3(256), 0, st>>>((const uint32_t*)bins, L.padded_bins, x8, first_slot, n_slots, (uint2*)mb, mb_n, mb_big, pitch, flags);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

// rows of queries one pass serves for a block of n_q: 32, 64, 128 or 256
uint32_t msc_pair_gemm_rows(uint32_t n_q) { return n_q <= 32 ? 32 : n_q <= 64 ? 64 : n_q <= 128 ? 128 : 256; }

uint32_t msc_pair_gemm_slices(uint64_t nbins, uint32_t m, uint32_t qn, int num_cus) {
	// enough workgroups for a few rounds of the chip (3 workgroups of 128 candidates per CU); a slice is an even number of 128-bin steps
	uint32_t s = 1;
	auto can_split = [&] { return s < 64 && nbins / (2 * s) >= 2 * kStep && nbins % (2 * s * 2 * kStep) == 0; };
	while (can_split() && (uint64_t)((m + 127) / 128) * s < (uint64_t)num_cus * 12) s *= 2;
	// and slices short enough that the queries' image of ONE slice stays in an XCD's 4 MiB L2 while the workgroups of that slice --
	// dispatched together -- walk it in step (msc_dot_gemm.hip measured 8 slices of 2 MiB best with 64 rows)
	static const uint64_t a_bytes = [] { const char* e = getenv("MSC_GEMM_A_KIB"); return (uint64_t)(e ? std::max(64, atoi(e)) : 4096) << 10; }();
	while (can_split() && (uint64_t)qn * (nbins / s) > a_bytes) s *= 2;
	return s;
}

// bytes of the queries' side of a block: the flag image and the transposed bytes
uint64_t msc_pair_gemm_image_bytes(uint64_t nbins, uint32_t qn) { return nbins * qn; }

// The queries' side: fimg and qT (nbins x qn bytes each) of rows q_slots[0 .. n_q) of the mirror q_x8; when n_hot > 0 also the hot list
// (hot: n_hot entries; hot_ptr, hot_cursor, hot_cnt: nbins / 128 + 1 words each) from the queries' lists of large bins.
hipError_t msc_launch_pair_gemm_queries(hipStream_t st, uint64_t nbins, const uint8_t* q_x8, const void* q_mb, const uint32_t* q_mb_n, uint32_t q_pitch,
                                        const uint32_t* q_slots_dev, uint32_t n_q, uint32_t qn, uint8_t* fimg, uint8_t* qT, uint64_t n_hot, void* hot, uint32_t* hot_ptr,
                                        uint32_t* hot_cursor, uint32_t* hot_cnt) {
	if (n_q == 0 || n_q > qn || nbins % kStep) return hipErrorInvalidValue;
	const uint32_t nsteps = (uint32_t)(nbins / kStep);
	if (qn == 32) k_x8_gather<32><<<dim3(nsteps), dim3(256), 0, st>>>(q_x8, q_slots_dev, n_q, nbins, fimg, qT);
	else if (qn == 64) k_x8_gather<64><<<dim3(nsteps), dim3(256), 0, st>>>(q_x8, q_slots_dev, n_q, nbins, fimg, qT);
	else if (qn == 128) k_x8_gather<128><<<dim3(nsteps), dim3(256), 0, st>>>(q_x8, q_slots_dev, n_q, nbins, fimg, qT);
	else if (qn == 256) k_x8_gather<256><<<dim3(nsteps), dim3(256), 0, st>>>(q_x8, q_slots_dev, n_q, nbins, fimg, qT);
	else return hipErrorInvalidValue;
	hipError_t e = hipGetLastError();
	if (e != hipSuccess || n_hot == 0) return e;
	if ((e = hipMemsetAsync(hot_cnt, 0, (nsteps + 1) * sizeof(uint32_t), st)) != hipSuccess) return e;
	k_hot_count<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, hot_cnt);
	k_hot_scan<<<dim3(1), dim3(1024), 0, st>>>(hot_cnt, nsteps, hot_ptr, hot_cursor);
	k_hot_fill<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, hot_cursor, (uint2*)hot);
	return hipGetLastError();
}

// P1 [k_slices][m][qn] and, with a hot list, P2 [m][qn] (zeroed here) of the block's queries against m candidates (slot list, or slots
// first .. first + m - 1) of the mirror cand_x8
hipError_t msc_launch_pair_gemm(hipStream_t st, uint64_t nbins, const uint8_t* cand_x8, const uint32_t* cand_slots, uint64_t first, uint32_t m, const uint8_t* fimg,
                                uint32_t qn, uint32_t k_slices, const uint32_t* hot_ptr, const void* hot, int32_t* out_min, int32_t* out_diff) {
	if (m == 0) return hipSuccess;
	if (k_slices == 0 || nbins % ((uint64_t)k_slices * 2 * kStep)) return hipErrorInvalidValue;
	if (hot_ptr) {
		const hipError_t e = hipMemsetAsync(out_diff, 0, (size_t)m * qn * sizeof(int32_t), st);
		if (e != hipSuccess) return e;
	}
	const dim3 grid((m + 127) / 128, k_slices);
#define MSC_PG_GO(NRB) k_pair_gemm_x8<NRB><<<grid, dim3(256), 0, st>>>(cand_x8, cand_slots, first, m, fimg, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff)
	if (qn == 32) MSC_PG_GO(1);
	else if (qn == 64) MSC_PG_GO(2);
	else if (qn == 128) MSC_PG_GO(4);
	else if (qn == 256) MSC_PG_GO(8);
	else return hipErrorInvalidValue;
#undef MSC_PG_GO
	return hipGetLastError();
}
