// msc_layout.h -- HBM layout of a histogram slot, shared by host code and kernels.
//
// A histogram of N = 4^k bins of T is stored as S tiles. A tile is what ONE wavefront consumes per
// step of the pair kernel: 64 lanes x LPT coalesced 16-byte loads. Inside a tile the bins are
// permuted so that after those LPT loads lane l holds R = LPT*E LOGICALLY CONSECUTIVE bins
// (E = 16/sizeof(T)): the earth-mover's statistic (predict/Feature.cpp:1505-1518) is a running
// prefix over bin order, and with this layout each lane scans its own run in registers and one
// wave-level DPP scan per tile stitches the 64 runs together -- no LDS, no uncoalesced access.
//
//   logical bin e of a tile:  lane = e / R, r = e % R, load t = r / E, j = r % E
//   physical element index :  t*(64*E) + lane*E + j
//
// Every other statistic is order-independent, so the permutation is invisible to it.
// Histograms smaller than one 1 KiB tile (k <= 3 for u32 ...) are padded with zero bins at the
// logical end; kernels mask them out of the prefix statistic.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)      // hipcc builds the kernels; plain g++ (host drivers, tests) sees the same structs
#define MSC_HD __host__ __device__ __forceinline__
#else
#define MSC_HD inline
#endif

struct MscLayout {
	uint32_t esz;         // bytes per bin
	uint32_t E;           // bins per 16-byte load
	uint32_t LPT;         // 16-byte loads per lane per tile (1, 2 or 4)
	uint32_t R;           // bins per lane per tile
	uint32_t tile_bins;   // 64 * R
	uint32_t S;           // tiles per histogram
	uint64_t nbins;       // 4^k
	uint64_t padded_bins; // S * tile_bins
	uint64_t slot_bytes;  // padded_bins * esz
};

MSC_HD MscLayout msc_make_layout(int k, int dtype) {
	MscLayout L;
	L.esz = (uint32_t)dtype / 8;
	L.E = 16 / L.esz;
	L.nbins = 1ull << (2 * k);
	uint64_t hist_bytes = L.nbins * L.esz;
	L.LPT = hist_bytes >= 4096 ? 4 : (hist_bytes >= 2048 ? 2 : 1);
	L.R = L.LPT * L.E;
	L.tile_bins = 64 * L.R;
	L.S = (uint32_t)((L.nbins + L.tile_bins - 1) / L.tile_bins);
	L.padded_bins = (uint64_t)L.S * L.tile_bins;
	L.slot_bytes = L.padded_bins * L.esz;
	return L;
}

// logical bin -> physical element index inside the slot
MSC_HD uint64_t msc_phys_index(uint64_t bin, uint32_t E, uint32_t R) {
	const uint32_t tile_bins = 64 * R;
	uint64_t tile = bin / tile_bins;
	uint32_t e = (uint32_t)(bin % tile_bins);
	uint32_t lane = e / R, r = e % R, t = r / E, j = r % E;
	return tile * tile_bins + (uint64_t)t * (64 * E) + lane * E + j;
}

// Per-slot scalar record, followed in memory by S uint64 tile prefixes (exclusive sum of the bins of
// all earlier tiles). One record per slot, stride = msc_scalar_stride(S).
struct MscSlotScalars {
	uint64_t mag;          // DivergencePoint::mag as the reference would hold it
	uint64_t length;
	uint64_t sum;          // true sum of bins
	uint64_t sum_sq;
	uint64_t max_count;
	uint64_t one_mers[4];
	double   stddev;
	uint64_t overflow;
	uint64_t id;
	uint64_t n_kmers;      // k-mers counted into this slot by hist_build (diagnostic)
	uint64_t reserved[3];
};
static_assert(sizeof(MscSlotScalars) == 128, "scalar record is 128 bytes");

MSC_HD uint64_t msc_scalar_stride(uint32_t S) {
	return (sizeof(MscSlotScalars) + 8ull * S + 127) / 128 * 128;
}
