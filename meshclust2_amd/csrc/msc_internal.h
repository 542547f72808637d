// msc_internal.h -- declarations shared between the C-ABI translation unit and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/meshclust2_hip.h"
#include "msc_layout.h"

// ---------------------------------------------------------------- sparse slots (sparse.hip)
#define MSC_SPARSE_SUB 16
struct MscSparseHdr {
	uint64_t off;                       // first entry of this slot in the set's entry arena
	uint32_t nnz;                       // entries (bins with value >= 2)
	uint32_t split[MSC_SPARSE_SUB + 1]; // entry offsets (relative to off) of the 16 equal index sub-ranges; split[16] == nnz
	uint32_t pad_[2];
};
static_assert(sizeof(MscSparseHdr) == 88, "sparse header layout");

// ---------------------------------------------------------------- model, as the epilogue kernel sees it
struct MscDevModel {
	int32_t  n_singles;
	int32_t  n_combos;
	uint64_t single_flag[MSC_MAX_SINGLES];
	double   mins[MSC_MAX_SINGLES];
	double   maxs[MSC_MAX_SINGLES];
	int32_t  is_sim[MSC_MAX_SINGLES];
	int32_t  combo_kind[MSC_MAX_COMBOS];
	int32_t  combo_n[MSC_MAX_COMBOS];
	int32_t  combo_idx[MSC_MAX_COMBOS][2];
	double   weights[MSC_MAX_COMBOS + 1];
	double   bias;
	// f32 image of the same model for the close-flag screen of k_pair_epilogue_bits (pair_features.hip, screen_close): filled by
	// msc_model_create; screen_ok = 0 when the model is outside what the screen bounds (a statistic that is not an integer reduction,
	// a bias, a non-finite constant)
	int32_t  screen_ok;
	float    s_min[MSC_MAX_SINGLES];          // mins[i]
	float    s_inv[MSC_MAX_SINGLES];          // 1 / (maxs[i] - mins[i])
	float    s_w[MSC_MAX_COMBOS + 1];         // weights
};

// One 1 x M problem of a batched launch (k_pair_tiles_batch, k_colsum_batch): query slot, its candidates' positions in the
// concatenated slot list, and the length window of Trainer::filter (ignored when the launch does not use a window).
struct MscBatchSeg {
	uint32_t q_slot, first, m, pad_;
	uint64_t min_len, max_len;
};

// One partial record per (candidate, tile): integer reductions of the streaming pass.
struct MscPartial {
	uint64_t manh;   // sum |p - q|
	uint64_t dot;    // sum p * q
	uint64_t emd;    // sum |prefix(p) - prefix(q)|
};

// Per-candidate result of the epilogue kernel.
struct MscPairOut {
	double  sum;       // weighted sum s
	double  csum;      // logistic(s) + bias
	double  combo0;    // first combo ("similarity" used for the arg-max)
	int32_t status;    // 0 ok, 1 skipped by the length window, MSC_ERR_* (negative) if the reference would throw
	int32_t close;     // round(csum) > 0
};

struct MscReduceOut {
	double  best_sim;
	int64_t best_pos;
	int32_t any_close;
	int32_t first_error;   // most negative status seen (0 if none)
	uint64_t n_close;
};

enum { MSC_REDUCE_GET_CLOSE = 0, MSC_REDUCE_MERGE = 1 };

// epilogue request
struct MscEpilogueArgs {
	const MscPartial* partials;       // [m][S]
	const void* partials16;           // or: [m][S] records of four u32 {manh, dot, emd, 0} (ring kernel); partials is then unused
	const void* partials_cq;          // or: [m_per_query][ceil(n_queries/16)][S][16] such records (digest kernel: one 256-byte run per workgroup step)
	uint32_t cq_group;                // with partials_cq: queries per group -- 16, or 32 with 4-byte records (manh only: dot_gemm set, emd from emd_ranks or not wanted)
	const uint64_t* emd_ranks;        // with partials_cq: the earth mover's distances from msc_emd_ranks.hip instead, [m_per_query][64] (query q at [q])
	const int32_t* dot_gemm;          // with partials_cq: the products from msc_dot_gemm.hip instead, [dot_slices][m_per_query][dot_stride] (query q at [q]);
	                                  // the records of partials_cq are then 8 bytes (manh, emd)
	uint32_t dot_slices, dot_stride;
	// the r04 pass on the matrix cores (msc_pair_gemm.hip, k_pair_epilogue_bits): P1 per slice and P2 of the block, the lists of large
	// bins (e = count - 1 >= 2) of both sets, the queries' counts [bin][kb_qn] (bytes, clamped at 127)
	const int32_t* kb_min;            // [kb_slices][m_per_query][kb_qn]; non-null selects k_pair_epilogue_bits
	const int32_t* kb_diff;           // [m_per_query][kb_qn], or null when no query of the block has a large bin
	uint32_t kb_slices, kb_qn;
	uint64_t kb_first;                // slot of candidate 0 when cand_slots is null (cand_scalars is then already offset to it)
	const void* kb_c_mb;              // uint2 (bin, e) [slot][kb_c_pitch]
	const uint32_t* kb_c_mb_n;
	uint32_t kb_c_pitch;
	const void* kb_q_mb;
	const uint32_t* kb_q_mb_n;
	uint32_t kb_q_pitch;
	const uint8_t* kb_qT;
	uint32_t emd_stride;              // entries per candidate of emd_ranks (0 = 64)
	const void* div_partials;         // [m][S] {jd, js} doubles, or null when no divergence statistic is requested
	uint32_t S;
	uint32_t m;
	const uint8_t* cand_scalars;      // scalar records of the candidate set
	uint64_t cand_scalar_stride;
	const uint32_t* cand_slots;       // nullable -> identity
	const uint8_t* q_scalars;         // scalar record of the query (already offset to its slot)
	// several queries in one launch (virtual candidate index = qi * m_per_query + ci); n_queries <= 1 -> q_scalars above
	uint32_t n_queries, m_per_query;
	const uint32_t* q_slots;          // device, [n_queries]
	const uint8_t* qset_scalars;
	uint64_t q_scalar_stride;
	uint64_t nbins;
	int32_t  dtype;
	int32_t  order;                   // MSC_ORDER_*
	int32_t  use_window;              // skip candidates outside [min_len, max_len]
	uint64_t min_len, max_len;
	// raw feature output
	uint64_t feat_mask;               // bits to emit in ascending order
	double*  raw_out;                 // [m][popcount(feat_mask)] or null
	// model output
	const MscDevModel* model;         // device pointer or null
	double*  singles_out;             // [m][n_singles] or null
	double*  combos_out;              // [m][n_combos] or null
	MscPairOut* pair_out;             // [m] or null
	// structure-of-arrays outputs (device, nullable): what the all-pairs path copies back instead of pair_out
	double*  sum_soa;
	double*  csum_soa;
	uint8_t* close_soa;
	int32_t  screen;                  // k_pair_epilogue_bits: only close_soa is wanted and the model has an f32 image -- decide in f32 with an error bound, FP64 where that does not decide
	int32_t* error_word;              // atomicMin of negative statuses
	// batched launch (k_pair_tiles_batch): candidate c belongs to segment pair_seg[c]; query slot and window come from it
	const MscBatchSeg* segs;
	const uint32_t* pair_seg;
	uint64_t sparse_base;             // 4^k when the partials come from k_pair_sparse (sums over the union only), else 0
	// divergence statistics from the sparse merge kernels (sparse sets, and dense sets through their sparse mirror): per pair
	// div_direct_n records of {jd, js} summed over the union of stored bins, relative to the (1, 1) term; div_base = 4^k adds it back
	const double* div_direct;         // [pairs][div_direct_n][2], pair index = the virtual candidate index
	uint32_t div_direct_n;
	uint64_t div_base;
	// statistics over 4-bin groups (sim_mm through markov, rre_k_r; k_pair_sparse_groups / k_sparse_self_markov): 16 partial
	// records per pair / per histogram, summed in sub-range order by the epilogue
	const double* grp_pairs;          // [pairs][16][2] = {markov, rre}
	const double* grp_self_c;         // [m_per_query or m][16]: markov(c, c) of every candidate of the launch (index = candidate position)
	const double* grp_self_q;         // [n_queries or 1][16]: markov(q, q)
};

// ---------------------------------------------------------------- launchers (defined in the .hip kernel files)
hipError_t msc_launch_fill(hipStream_t st, void* bins, const MscLayout& L, uint64_t first_slot, uint64_t n_slots);
hipError_t msc_launch_count(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype,
                            uint64_t first_slot, const uint32_t* packed_words, const uint32_t* seg_seq,
                            const uint64_t* seg_start, const uint64_t* kmer_off, uint64_t n_segs,
                            uint64_t total_kmers, bool saturating);
hipError_t msc_launch_finalize(hipStream_t st, const void* bins, uint8_t* scalars, const MscLayout& L, int dtype,
                               uint64_t first_slot, uint64_t n_slots, bool keep_mag, uint64_t* tile_scratch = nullptr);
bool msc_lds_build_supported(const MscLayout& L);
bool msc_sort_build_supported(const MscLayout& L, int k);
uint32_t msc_sort_build_max_kmers();
hipError_t msc_launch_build_sort(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype, uint64_t first_slot,
                                 const uint32_t* seq_ids, uint64_t n, uint32_t P, const uint32_t* packed_words, const uint64_t* seg_start,
                                 const uint64_t* kmer_off, const uint64_t* seq_seg_begin, const uint64_t* seq_meta, uint64_t* bounds);
hipError_t msc_launch_build_lds(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype, uint64_t first_slot,
                                uint64_t n_seqs, const uint32_t* packed_words, const uint64_t* seg_start, const uint64_t* kmer_off,
                                const uint64_t* seq_seg_begin);
hipError_t msc_launch_permute(hipStream_t st, const void* src, void* dst, const MscLayout& L, int dtype, bool to_physical);

hipError_t msc_launch_pair_tiles(hipStream_t st, const MscLayout& L, int dtype,
                                 const uint8_t* cand_bins, const uint8_t* cand_scalars, const uint32_t* cand_slots,
                                 uint32_t m, const uint8_t* q_bins_slot, const uint8_t* q_scalars_slot,
                                 int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials,
                                 int num_cus, void* div_tables /*nullable*/, void* div_partials, int order);
hipError_t msc_launch_pair_tiles_multi(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                       const uint32_t* cand_slots, uint32_t m, const uint8_t* qset_bins, uint64_t q_slot_bytes,
                                       const uint8_t* qset_scalars, uint64_t q_scalar_stride, const uint32_t* q_slots, uint32_t n_q,
                                       int tq, bool compact, MscPartial* partials, int num_cus);
hipError_t msc_launch_pair_tiles_wide(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                      const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins_slot, const uint8_t* q_scalars_slot,
                                      int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus,
                                      void* div_partials /*nullable*/, int order);
int msc_div_table_dim(const MscLayout& L);      // 8 or 16: side of the per-candidate (count, count) term table
// Q x M streaming through the LDS-DMA ring (32/64-bit bins, LPT 4, compact totals, tq 4 or 8). hipErrorInvalidValue = not applicable.
hipError_t msc_launch_pair_tiles_multi_ring(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                            const uint32_t* cand_slots, uint32_t m, const uint8_t* qset_bins, uint64_t q_slot_bytes,
                                            const uint8_t* qset_scalars, uint64_t q_scalar_stride, const uint32_t* q_slots, uint32_t n_q, int tq,
                                            bool prefix16, void* partials16, int num_cus);
// digest mirror of an 8/16/32-bit set (pair_digest.hip): 4 bytes per bin whatever the bin type; build for slots [first, first+n),
// and the Q x M pass that streams it
bool msc_digest_supported(const MscLayout& L);
uint64_t msc_digest_slot_bytes(const MscLayout& L);
hipError_t msc_launch_digest_build(hipStream_t st, const MscLayout& L, const uint8_t* bins, const uint8_t* scalars, uint8_t* digest, uint64_t first_slot,
                                   uint64_t n_slots);
int msc_digest_tiles_per_step(const MscLayout& L, uint64_t max_count);      // 1 or 2 consecutive tiles scored per loop step
hipError_t msc_launch_pair_digest_multi(hipStream_t st, const MscLayout& L, const uint8_t* cand_digest, const uint32_t* cand_slots, uint32_t m,
                                        const uint8_t* q_digest, const uint32_t* q_slots, uint32_t n_q, bool counts_fit_u8,
                                        int tiles_per_step, bool need_emd, void* partials16, int num_cus, bool need_dot = true, int queries_per_wave = 4);
// the products of the Q x M pass as an int8 GEMM on the matrix cores (msc_dot_gemm.hip)
uint64_t msc_ranks_pitch(uint64_t max_excess);
hipError_t msc_launch_ranks_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint8_t* scalars, uint32_t* ranks, uint32_t* n_of, uint64_t pitch,
                                  uint64_t first_slot, uint64_t n_slots, int32_t* bad);
hipError_t msc_launch_emd_ranks(hipStream_t st, uint64_t nbins, const uint32_t* c_ranks, uint64_t c_pitch, const uint32_t* c_n, const uint32_t* cand_slots, uint64_t first,
                                uint32_t m, const uint32_t* q_ranks, uint64_t q_pitch, const uint32_t* q_n, const uint32_t* q_slots_dev, uint32_t n_q, uint64_t* out,
                                uint32_t out_stride = 64);
hipError_t msc_launch_ranks16_build(hipStream_t st, uint64_t nbins, const uint32_t* ranks, uint16_t* ranks16, uint64_t pitch, uint64_t first_slot, uint64_t n_slots, int32_t* bad);
hipError_t msc_launch_emd_ranks16(hipStream_t st, uint64_t nbins, const uint16_t* c_ranks, uint64_t pitch, const uint32_t* c_n, const uint32_t* cand_slots, uint64_t first,
                                  uint32_t m, const uint16_t* q_ranks, const uint32_t* q_n, const uint32_t* q_slots_dev, uint32_t n_q, uint64_t* out, uint32_t out_stride);
// the r04 form of that pass (msc_pair_gemm.hip): presence-bit mirror + lists of large bins, the queries' side of a block, the product
uint64_t msc_kb_bytes(const MscLayout& L, uint64_t capacity);
hipError_t msc_launch_kb_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, uint8_t* kb, uint64_t first_slot, uint64_t n_slots, void* mb,
                               uint32_t* mb_n, uint32_t pitch, int32_t* flags);
uint32_t msc_pair_gemm_rows(uint32_t n_q);
// msc_ranks_pass.hip: the 1 x M pass over rank lists
size_t msc_ranks_pass_lds(uint64_t nbins, uint64_t q_kmers);
uint32_t msc_ranks_pass_query_cap();
hipError_t msc_launch_rank_lists_sizes(hipStream_t st, const MscSparseHdr* hdr, const uint32_t* cum, uint64_t capacity, uint32_t* n, uint64_t* off);
hipError_t msc_launch_rank_lists_fill(hipStream_t st, const void* ent, const uint32_t* cum, const MscSparseHdr* hdr, uint64_t capacity, const uint32_t* n, const uint64_t* off,
                                      uint64_t nbins, uint32_t* out);
// the divergence statistics of a long-list rank pass (msc_ranks_pass.hip): what the launch needs beside its own scratch
struct MscRankDiv {
	uint32_t* big;              // [the query's stored bins] the counts >= 8
	const uint8_t* q_scalars;   // the query's scalar record
	int order;
	double* div_out;            // [m][2] the {jd, js} record of each candidate
};
hipError_t msc_launch_pair_ranks_1xm(hipStream_t st, const uint32_t* c_rk, const uint64_t* c_off, const uint32_t* c_n, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                     const uint32_t* cand_slots, uint64_t first, uint32_t m, const void* q_ent, const uint32_t* q_cum, const MscSparseHdr* q_hdr, uint64_t nbins,
                                     int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus, uint64_t q_kmers, uint32_t* guard, uint32_t* q_scratch);
uint64_t msc_ranks_pass_query_scratch(uint64_t q_kmers);
uint32_t msc_ranks_items_round();
size_t msc_ranks_items_rec_bytes(uint64_t m, uint32_t rounds);                   // the pass's records and spot-term slots, one per item
size_t msc_ranks_items_list_bytes(uint64_t m, uint32_t rounds);                  // ... its candidates' records and its list of items
hipError_t msc_launch_pair_ranks_items(hipStream_t st, const uint32_t* c_rk, const uint64_t* c_off, const uint32_t* c_n, const void* c_rm, const uint64_t* c_rm_off,
                                       const uint32_t* c_rm_n, const uint8_t* cand_scalars, uint64_t scalar_stride, const uint32_t* cand_slots, uint64_t first, uint32_t m,
                                       const void* q_ent, const uint32_t* q_cum, const MscSparseHdr* q_hdr, uint64_t nbins, int use_window, uint64_t min_len, uint64_t max_len,
                                       MscPartial* partials, int num_cus, uint32_t* q_scratch, uint32_t rounds, void* rec_scratch, const MscRankDiv* dv, uint64_t q_kmers,
                                       uint32_t* guard, void* item_scratch, uint32_t* counters, uint32_t* tables, int turn, const uint32_t* q_rk);
uint32_t msc_ranks_items_table_words(uint64_t nbins);
// the repeated-bin lists of a sparse set (bin, count - 1 for the bins counted twice and more), beside its rank lists
hipError_t msc_launch_rank_multi_sizes(hipStream_t st, const void* ent, const MscSparseHdr* hdr, uint64_t capacity, uint32_t* n, uint64_t* off);
hipError_t msc_launch_rank_multi_fill(hipStream_t st, const void* ent, const MscSparseHdr* hdr, uint64_t capacity, const uint64_t* off, void* out);
const char* msc_pair_gemm_kernel_name();          // "k_pair_gemm_fp4_dma"
uint32_t msc_pair_gemm_slices(uint64_t nbins, uint32_t m, uint32_t qn, int num_cus);
uint64_t msc_pair_gemm_qt_bytes(uint64_t nbins, uint32_t qn);
uint64_t msc_pair_gemm_anib_bytes(uint64_t nbins, uint32_t qn);          // the queries' nibble tiles for the LDS-DMA form of the product (0: off)
hipError_t msc_launch_pair_gemm_queries(hipStream_t st, uint64_t nbins, const uint8_t* q_kb, const void* q_mb, const uint32_t* q_mb_n, uint32_t q_pitch,
                                        const uint32_t* q_slots_dev, uint32_t n_q, uint32_t qn, uint8_t* qT, uint64_t n_hot, void* hot, uint32_t* hot_ptr,
                                        uint32_t* hot_cursor, uint32_t* hot_cnt, uint8_t* anib);
hipError_t msc_launch_pair_gemm(hipStream_t st, uint64_t nbins, const uint8_t* cand_kb, const uint32_t* cand_slots, uint64_t first, uint32_t m, uint32_t qn,
                                uint32_t k_slices, const uint32_t* hot_ptr, const void* hot, int32_t* out_min, int32_t* out_diff, const uint8_t* anib);
hipError_t msc_launch_epilogue(hipStream_t st, const MscEpilogueArgs& a);
hipError_t msc_launch_close_counts(hipStream_t st, const uint8_t* flags, uint32_t n_q, uint32_t m, uint64_t* counts);
// the window bookkeeping the fused epilogue + reduce kernels do for msc_get_close_window (msc_window.hip): pos[i] = position of candidate i,
// alive[] = the window's flags, counter / out = the host-visible list of closed positions (out[0] = best position + 1, out[2 ..] = the list)
struct MscCloseList {
	const uint32_t* pos;
	uint8_t* alive;
	uint32_t* counter;
	uint32_t* out;
	uint32_t* out_host = nullptr;      // the host's own pointer to `out` (page-locked memory the device addresses)
};
// what a workgroup of the fused epilogue + reduce kernel leaves of its candidates: its best (index and, in a window pass, position + 1), its count of
// close ones, its first error. Folded by k_pair_reduce_fold2 -- or, host_parts != nullptr (page-locked, device-addressable), by the caller
// once it has waited for the stream: a launch less per pass of the step-serial loop.
struct ReducePart { double sim; int64_t pos; unsigned long long nclose; int err; uint32_t wpos; };
hipError_t msc_launch_epilogue_reduce(hipStream_t st, const MscEpilogueArgs& a, int mode, int64_t begin, uint8_t* flags_out, MscReduceOut* out, void* parts_scratch,
                                      const MscCloseList& cl, ReducePart* host_parts = nullptr, uint32_t* n_parts_out = nullptr);
// ... the fold of those parts on the host (the arithmetic of k_pair_reduce_fold2); *wpos_out = the best candidate's window position + 1 (0: none)
void msc_reduce_fold_host(const ReducePart* parts, uint32_t n_parts, int mode, MscReduceOut* out, uint32_t* wpos_out);
hipError_t msc_launch_reduce(hipStream_t st, const MscPairOut* pair_out, uint32_t m, int mode, int64_t begin,
                             uint8_t* flags_out, MscReduceOut* out, void* parts_scratch = nullptr, const uint32_t* key = nullptr);
size_t msc_reduce_scratch_bytes();

hipError_t msc_launch_sparse_count(hipStream_t st, const void* scratch_bins, const MscLayout& L, int dtype, uint32_t n, uint64_t* counts);
hipError_t msc_launch_sparse_write(hipStream_t st, const void* scratch_bins, const MscLayout& L, int dtype, uint32_t n, const MscSparseHdr* hdr,
                                   uint64_t first_slot, const uint64_t* cum_base, void* ent, uint32_t* cum);
hipError_t msc_launch_pair_sparse(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                  uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                  const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, int use_window, uint64_t min_len,
                                  uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order);
hipError_t msc_launch_pair_sparse_lds(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                      uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                      const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, uint32_t q_nnz, uint32_t max_c_nnz, int use_window,
                                      uint64_t min_len, uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, int num_cus);
uint32_t msc_sparse_mp_max_entries();
bool msc_sparse_wl_fits(uint32_t q_nnz, uint32_t c_max_nnz);      // the whole-list kernel takes this 1 x M pass (one record per candidate)
uint32_t msc_sparse_mp_parts(uint32_t m, uint64_t entries, int num_cus, bool div = false);      // waves per candidate for a window of m (1 .. 16)
uint32_t msc_sparse_mp_div_records(uint64_t entries);      // {jd, js} records per pair the merge-path kernel's divergence form writes
hipError_t msc_launch_pair_sparse_mp(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                     uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                     const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, int use_window, uint64_t min_len,
                                     uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, int num_cus, uint32_t max_total, uint32_t parts, uint32_t q_nnz = 0, uint32_t c_max_nnz = 0,
                                     uint32_t div_stride = 1);
hipError_t msc_launch_sparse_build_sort(hipStream_t st, int k, int dtype, uint64_t nbins, uint64_t first_slot, uint32_t n_seqs, const uint32_t* packed,
                                        const uint64_t* seg_start, const uint64_t* kmer_off, const uint64_t* seq_seg_begin, const uint64_t* seq_arena_off,
                                        uint32_t P, uint8_t* scalars, uint64_t scalar_stride, MscSparseHdr* hdr, void* ent, uint32_t* cum);
hipError_t msc_launch_sparse_scatter(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, uint32_t m, uint32_t* acc);
hipError_t msc_launch_sparse_mean_count(hipStream_t st, int dtype, const uint32_t* acc, uint32_t n_chunks, uint64_t chunk_bins, uint32_t m, uint64_t* counts);
hipError_t msc_launch_sparse_mean_write(hipStream_t st, int dtype, uint32_t* acc, uint32_t n_chunks, uint64_t chunk_bins, uint32_t m,
                                        const uint64_t* chunk_off, const uint64_t* chunk_cum, void* ent, uint32_t* cum);
hipError_t msc_launch_pair_sparse_mp_pairs(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                           uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                           const MscSparseHdr* q_hdr, uint64_t nbins, int use_window, const MscBatchSeg* segs, const uint32_t* pair_seg,
                                           MscPartial* partials, int order, int num_cus, const uint8_t* q_scalars = nullptr, uint64_t q_scalar_stride = 0,
                                           void* div_tables = nullptr, void* div_partials = nullptr, uint32_t div_stride = 1);
hipError_t msc_launch_sparse_scatter_batch(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, const uint32_t* seg, uint32_t n_members,
                                           uint64_t nbins, uint32_t* acc, uint32_t* touched = nullptr);
hipError_t msc_launch_sparse_mean_count_batch(hipStream_t st, int dtype, const uint32_t* acc, uint64_t nbins, uint32_t n_chunks, uint64_t chunk_bins, uint32_t n_centres,
                                              const uint32_t* m_of, uint64_t* counts, const uint32_t* touched = nullptr);
hipError_t msc_launch_sparse_mean_write_batch(hipStream_t st, int dtype, uint32_t* acc, uint64_t nbins, uint32_t n_chunks, uint64_t chunk_bins, uint32_t n_centres,
                                              const uint32_t* m_of, const uint64_t* chunk_off, const uint64_t* chunk_cum, void* ent, uint32_t* cum, uint32_t* touched = nullptr);
hipError_t msc_launch_sparse_assign_batch(hipStream_t st, void* d_ent, uint32_t* d_cum, MscSparseHdr* d_hdr, const void* s_ent, const uint32_t* s_cum, const MscSparseHdr* s_hdr,
                                          const uint32_t* ds, const uint32_t* ss, const uint64_t* dst_off, uint32_t n);
hipError_t msc_launch_assign_scalars(hipStream_t st, uint8_t* dst_scalars, const uint8_t* src_scalars, uint64_t stride_bytes, const uint32_t* dst_slots,
                                     const uint32_t* src_slots, uint32_t n, int exact = 0);
hipError_t msc_launch_pair_sparse_groups(hipStream_t st, const void* c_ent, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                         const uint32_t* cand_slots, uint32_t m, const void* q_ent, const MscSparseHdr* q_hdr, int use_window, uint64_t min_len,
                                         uint64_t max_len, double* out);
hipError_t msc_launch_sparse_nnz_sum(hipStream_t st, const MscSparseHdr* hdr, const uint8_t* cand_scalars, uint64_t scalar_stride, const uint32_t* slots, uint64_t first_slot,
                                     uint32_t m, int use_window, uint64_t min_len, uint64_t max_len, uint64_t* acc);
hipError_t msc_launch_sparse_self_markov(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, uint64_t first_slot, uint32_t m, double* out);
// the same records from tile-permuted dense slots, for histograms under 64 KiB (pair_features.hip)
hipError_t msc_launch_pair_groups_dense(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* c_bins, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                        const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins, int use_window, uint64_t min_len, uint64_t max_len, double* out);
hipError_t msc_launch_self_markov_dense(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* slots, uint64_t first_slot, uint32_t m,
                                        double* out);
hipError_t msc_launch_colsum(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins,
                             const uint32_t* member_slots, uint32_t m, void* rounded_out /*T, physical*/,
                             double* mean_out /*physical, nullable*/, uint64_t* floor_sum_out, uint64_t* scratch);
hipError_t msc_launch_distance_d(hipStream_t st, const MscPartial* partials, uint32_t S, uint32_t m,
                                 const uint8_t* scalars, uint64_t scalar_stride, const uint32_t* member_slots,
                                 const uint8_t* r_scalars, const uint64_t* floor_sum, double* dist_out,
                                 MscReduceOut* out);
// batched forms for msc_update_centres: one (query, candidate list) problem per segment
hipError_t msc_launch_pair_tiles_batch(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                       const uint32_t* cand_slots, const MscBatchSeg* segs, uint32_t n_segs, uint32_t max_m, const uint8_t* qset_bins,
                                       uint64_t q_slot_bytes, const uint8_t* qset_scalars, uint64_t q_scalar_stride, int use_window, MscPartial* partials,
                                       int order);
hipError_t msc_launch_colsum_batch(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* member_slots, const MscBatchSeg* segs,
                                   uint32_t n_segs, void* rounded_out, uint64_t* floor_sum_out);
hipError_t msc_launch_colsum_sums(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* member_slots, const MscBatchSeg* segs, uint32_t n_segs,
                                  uint64_t* sums_out);
hipError_t msc_launch_mean_from_sums(hipStream_t st, const MscLayout& L, int dtype, const uint64_t* sums, const uint64_t* m_total, uint32_t n_segs, void* rounded_out,
                                     uint64_t* floor_sum_out);
hipError_t msc_launch_distance_batch(hipStream_t st, const MscPartial* partials, uint32_t S, uint32_t n, const uint8_t* scalars, uint64_t scalar_stride,
                                     const uint32_t* member_slots, const uint32_t* pair_seg, const uint8_t* r_scalars, uint64_t r_stride, const uint64_t* floor_sum,
                                     double* dist_out);
hipError_t msc_launch_assign_batch(hipStream_t st, const MscLayout& L, uint8_t* dst_bins, uint8_t* dst_scalars, const uint8_t* src_bins,
                                   const uint8_t* src_scalars, const uint32_t* dst_slots, const uint32_t* src_slots, uint32_t n, int exact = 0);
