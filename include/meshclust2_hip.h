/* include/meshclust2_hip.h
 *
 * C ABI of libmeshclust2_hip.so: the MI355X (gfx950) replacement for MeShClust2's alignment-free
 * pairwise-identity hot path. Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * The reference (BioinformaticsToolsmith/MeShClust2 v2.3.0) has no FFI layer: the path is reached through
 * C++ template member calls. Each entry point below names the reference call site(s) it replaces
 * (file:line relative to the reference's src/). INTEGRATION.md shows the shim a maintainer would add on
 * the reference side.
 *
 * Conventions
 *   - every function returns MSC_OK (0) or a negative msc_status; msc_last_error(ctx) has the message.
 *     Nothing throws across the ABI (the reference throws std::exception / const char* / int).
 *   - a msc_ctx owns one HIP device + one stream and is NOT thread-safe: use one ctx per host thread
 *     (the reference calls compute()/classify() from OpenMP workers on shared read-only state,
 *     cluster/Trainer.cpp:41,84).
 *   - calls are synchronous: results are valid when the call returns.
 *   - histograms live in HBM inside a msc_hist_set from the moment they are built; the host sees
 *     slot indices, flags and scalars. `dtype` is the reference's --datatype: 8, 16, 32 or 64
 *     (cluster/CRunner.cpp:108-126), fixed per set.
 *   - there is NO CPU fallback: msc_create() fails with MSC_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef MESHCLUST2_HIP_H
#define MESHCLUST2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSC_ABI_VERSION 1

typedef enum {
	MSC_OK = 0,
	MSC_ERR_INVALID_ARG = -1,
	MSC_ERR_NO_DEVICE = -2,       /* no usable gfx950 device / HIP runtime error at init */
	MSC_ERR_HIP = -3,             /* a HIP call failed; message has hipGetErrorString */
	MSC_ERR_OOM = -4,
	MSC_ERR_INVALID_INPUT = -5,   /* InvalidInputException: character outside the IUPAC map (nonltr/ChromosomeOneDigit.cpp:86-94) */
	MSC_ERR_ZERO_LENGTH = -6,     /* `throw 123`: length_difference on a zero-length point (predict/Feature.cpp:878-886) */
	MSC_ERR_NAN = -7,             /* normalize_cache found NaN (predict/Feature.cpp:143-146) */
	MSC_ERR_UNSUPPORTED = -8,     /* feature flag outside the 11 in-scope statistics, k out of range, ... */
	MSC_ERR_IO = -9
} msc_status;

/* Single-feature bits: identical values to FEAT_* in predict/Feature.h:31-64. */
#define MSC_FEAT_MANHATTAN          (1ULL << 2)
#define MSC_FEAT_EUCLIDEAN          (1ULL << 3)
#define MSC_FEAT_NORMALIZED_VECTORS (1ULL << 5)
#define MSC_FEAT_JEFFEREY_DIV       (1ULL << 7)
#define MSC_FEAT_PEARSON_COEFF      (1ULL << 9)
#define MSC_FEAT_INTERSECTION       (1ULL << 13)
#define MSC_FEAT_EMD                (1ULL << 18)
#define MSC_FEAT_LENGTHD            (1ULL << 21)
#define MSC_FEAT_KULCZYNSKI2        (1ULL << 27)
#define MSC_FEAT_SIMRATIO           (1ULL << 28)
#define MSC_FEAT_JENSEN_SHANNON     (1ULL << 29)
/* PRED_FEAT_FAST / PRED_FEAT_DIV, predict/Predictor.h:23-24 (`--feat fast` = FAST, `--feat slow` = FAST|DIV) */
#define MSC_FEAT_FAST (MSC_FEAT_EUCLIDEAN | MSC_FEAT_MANHATTAN | MSC_FEAT_INTERSECTION | MSC_FEAT_KULCZYNSKI2 | \
                       MSC_FEAT_SIMRATIO | MSC_FEAT_NORMALIZED_VECTORS | MSC_FEAT_PEARSON_COEFF | MSC_FEAT_EMD | MSC_FEAT_LENGTHD)
#define MSC_FEAT_DIV  (MSC_FEAT_JEFFEREY_DIV | MSC_FEAT_JENSEN_SHANNON)
#define MSC_FEAT_SLOW (MSC_FEAT_FAST | MSC_FEAT_DIV)
/* Two of the reference's `extraslow` statistics (PRED_FEAT_ALL, predict/Predictor.h:25): sums over the groups of four neighbouring
 * bins that share a (k-1)-mer prefix. rre_k_r: predict/Feature.cpp:1029-1062; sim_mm = 1 - exp((d_markov(a,b) + d_markov(b,a)) / 2)
 * with Feature<T>::markov, :1367-1393,1429-1455. Scored on sorted (bin, value) lists (sparse sets, and dense sets of histograms
 * >= 64 KiB through their sparse mirror) or, for smaller histograms, straight from the dense slots -- same terms, same order;
 * usable in msc_pair_features_raw, msc_score, the Trainer operators and models, not in msc_train_class. */
#define MSC_FEAT_RRE_K_R            (1ULL << 14)
#define MSC_FEAT_SIM_MM             (1ULL << 16)
#define MSC_FEAT_GROUPS (MSC_FEAT_RRE_K_R | MSC_FEAT_SIM_MM)

/* Combo codes as written in the weights file (predict/Predictor.cpp:96-110). */
#define MSC_COMBO_XY   0
#define MSC_COMBO_XY2  1
#define MSC_COMBO_X2Y  2
#define MSC_COMBO_X2Y2 3

#define MSC_MAX_SINGLES 16
#define MSC_MAX_COMBOS  8

/* Argument order of a scored pair. All 11 statistics are symmetric EXCEPT the reference's uint32_t
 * simratio (its wrapped difference), so the order is part of the contract:
 *   MSC_ORDER_CAND_FIRST : f(candidate, query)  -- Trainer::get_close, merge (cluster/Trainer.cpp:49,93)
 *   MSC_ORDER_QUERY_FIRST: f(query, candidate)  -- Trainer::filter -> classify(p, pt) (cluster/Trainer.cpp:133) */
#define MSC_ORDER_CAND_FIRST  0
#define MSC_ORDER_QUERY_FIRST 1

typedef struct msc_ctx msc_ctx;
typedef struct msc_hist_set msc_hist_set;
typedef struct msc_model msc_model;

/* Scalar members of one DivergencePoint<T> (clutil/DivergencePoint.h:81-87) plus derived sums the kernels use. */
typedef struct {
	uint64_t mag;          /* DivergencePoint::mag as the reference holds it (stale after hist_assign) */
	uint64_t length;       /* effective length (Chromosome::getEffectiveSize) */
	uint64_t sum;          /* true sum of the bins */
	uint64_t sum_sq;       /* sum of squared bins */
	uint64_t max_count;    /* largest bin */
	uint64_t one_mers[4];  /* k=1 table with pseudocount 1 (clutil/Loader.cpp:143-153) */
	double   stddev;       /* clutil/Loader.cpp:158-171 */
	int32_t  overflow;     /* 1 if some bin saturated at max(T) (nonltr/KmerHashTable.cpp:248-252) */
	int32_t  pad_;
	uint64_t id;
} msc_hist_info;

/* ------------------------------------------------------------------ context */
int         msc_abi_version(void);
/* device >= 0: HIP ordinal. Fails (MSC_ERR_NO_DEVICE) if HIP cannot initialise that device. */
int         msc_create(int device, msc_ctx** out);
void        msc_destroy(msc_ctx* ctx);
const char* msc_last_error(const msc_ctx* ctx);     /* ctx may be NULL: last error of a failed msc_create */
int         msc_device_name(const msc_ctx* ctx, char* buf, size_t cap);
int         msc_synchronize(msc_ctx* ctx);
/* Wall-clock of the dominant kernel (pair_tiles) of the LAST scoring call, from HIP events on the ctx stream (ms). */
int         msc_last_kernel_ms(const msc_ctx* ctx, float* pair_tiles_ms, float* total_ms);
/* The events behind msc_last_kernel_ms cost four records per 1 x M call: a step-serial caller (the accumulate loop issues one
 * Trainer::get_close per step) switches them off; msc_last_kernel_ms then fails with MSC_ERR_INVALID_ARG. Default: on. */
int         msc_set_kernel_timing(msc_ctx* ctx, int on);
/* A dense set whose histograms have a list form (64 KiB and more) keeps a sparse mirror of its slots, and its 1 x M passes
 * (Trainer::get_close / filter / merge, Feature::compute: cluster/Trainer.cpp:23-141) merge the two lists -- 8 bytes per counted k-mer
 * instead of 4^k bins per candidate, same integer reductions. on = 0 keeps those passes on the streaming kernel over the bins (the
 * 1 x M shape SURVEY 8(d) prices at 4^k sizeof(T) bytes per pair); results are identical either way. Default: on
 * (off at start with MSC_NO_MIRROR_1XM in the environment). */
int         msc_set_mirror_pass(msc_ctx* ctx, int on);
/* msc_score_multi queues the blocks of a call (128 queries x all candidates each; fastcar's outer loop, fastcar/FC_Runner.cpp:585-597) on
 * three streams: block i's product beside its rank walk, the epilogue of block i - 1 and the queries' side of block i + 1. on = 0 runs
 * every kernel of a block on ONE stream, block after block: same results, and per-kernel timings (msc_last_kernel_ms, rocprofv3) that are
 * not stretched by a neighbour -- what the roofline of the product kernel is measured on. Default: on (off at start with
 * MSC_GEMM_NO_PIPE in the environment). */
int         msc_set_block_pipe(msc_ctx* ctx, int on);
/* Number of streaming-kernel launches that pair_tiles_ms sums over (large calls are chunked). */
int         msc_last_kernel_launches(const msc_ctx* ctx);
/* Which streaming kernel the LAST scoring call ran (its name is copied to buf) and how many queries one HBM read of a
 * candidate tile served in it (1 for the 1 x M passes): the figure the algorithmic bytes of a Q x M launch divide by. */
int         msc_last_kernel_info(const msc_ctx* ctx, char* buf, size_t cap, int* queries_per_candidate_read);

/* ------------------------------------------------------------------ a1: sequence encoding (host byte work)
 * Replaces Chromosome::help / removeAmbiguous / mergeSegments / makeSegmentList + ChromosomeOneDigit::encode
 * (nonltr/Chromosome.cpp:130-154,263-385; nonltr/ChromosomeOneDigit.cpp:79-133).
 * codes_out[len]: 0..3 where the reference encodes, raw 'N' elsewhere. segs_out: inclusive [s,e] pairs. */
int msc_encode(const char* seq, size_t len, uint8_t* codes_out, int64_t* segs_out, size_t max_segs,
               size_t* n_segs, uint64_t* eff_len);

/* ------------------------------------------------------------------ a2-a4: histogram sets */
/* A set holds `capacity` slots of 4^k bins of `dtype` bits in HBM (tile-permuted layout). */
int      msc_hist_set_create(msc_ctx* ctx, int k, int dtype, uint64_t capacity, msc_hist_set** out);
void     msc_hist_set_destroy(msc_hist_set* set);
uint64_t msc_hist_set_capacity(const msc_hist_set* set);
int      msc_hist_set_k(const msc_hist_set* set);
int      msc_hist_set_dtype(const msc_hist_set* set);
uint64_t msc_hist_set_bytes(const msc_hist_set* set);              /* HBM footprint */

/* SPARSE layout: a slot stores only the bins above the pseudocount, as a sorted (bin, value) list.
 * Same semantics and the same results as a dense set for msc_hist_build*, msc_hist_download (expanded on the host),
 * msc_hist_info_get, clone / assign / copy, msc_pair_features_raw, msc_score(_multi), msc_get_close, msc_filter, msc_merge and
 * msc_search; it is what makes k = 11..15 possible (a dense k = 13 histogram is 64-512 MiB) and it reads 12 bytes per
 * stored bin instead of 4^k * dtype/8 per histogram. max_entries = total stored bins the set can hold (sum over slots, <= the
 * total number of k-mers). Needs 4^k * dtype/8 >= 64 KiB. msc_hist_upload and the mean_out of msc_mean_nearest are not available on sparse sets. */
int      msc_hist_set_create_sparse(msc_ctx* ctx, int k, int dtype, uint64_t capacity, uint64_t max_entries, msc_hist_set** out);
int      msc_hist_set_is_sparse(const msc_hist_set* set);
uint64_t msc_hist_set_entries(const msc_hist_set* set, uint64_t slot);     /* stored bins of one slot (a dense set: of its sparse mirror, 0 before it exists) */
/* Empties a sparse set: every slot back to "never written", the whole entry arena free again. The entry arena is append-only
 * (a slot that is assigned again leaves its old list behind), so a store of moving centres is compacted by copying its live slots
 * into a second set (msc_hist_copy_batch) and clearing the first for the next time -- no allocation in the loop. The reference has
 * no counterpart: its centres are heap objects (Center<T>, cluster/Center.h). Dense sets: MSC_ERR_UNSUPPORTED. */
int      msc_hist_set_clear(msc_ctx* ctx, msc_hist_set* set);

/* Replaces Loader<T>::get_point (clutil/Loader.cpp:112-179; callers cluster/CRunner.cpp:526,
 * predict/Predictor.cpp:799,858, fastcar/FC_Runner.cpp:501) for n_seqs sequences at once.
 *   seqs[i]/lens[i] : raw ASCII sequence i (FASTA body, no header, no newlines).
 *   strip_non_acgt  : 1 = the std::string overload (drops every char that is not upper-case ACGT first,
 *                     Loader.cpp:115-121); 0 = the ChromosomeOneDigit overload used by the clustering CLI.
 * Slots first_slot .. first_slot+n_seqs-1 are overwritten. Host encodes + packs to 2 bits/base, the GPU
 * counts. MSC_ERR_INVALID_INPUT if a sequence holds a character outside the IUPAC map. */
int msc_hist_build(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs,
                   const char* const* seqs, const uint64_t* lens, int strip_non_acgt);

/* Same, from already-encoded input (what a caller with its own FASTA reader hands over):
 *   packed        : 2-bit codes, base b at bits [2*(b%4), 2*(b%4)+1] of byte b/4, all sequences concatenated
 *   seg_seq/start/end : n_segs segments; seg_seq[j] = sequence ordinal, [start,end] inclusive GLOBAL base offsets
 *                    (the reference's segment list, nonltr/Chromosome.cpp:355-385); only k-mers fully inside a
 *                    segment are counted (clutil/Loader.cpp:53-54)
 *   eff_len[i], one_mers[4*i..] : per-sequence effective length and k=1 counts (pseudocount included) */
int msc_hist_build_packed(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs,
                          const uint8_t* packed, uint64_t n_bases,
                          const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_segs,
                          const uint64_t* eff_len, const uint64_t* one_mers);

/* The same with `packed` in THIS device's memory (every other array on the host, as above): the form a multi-GPU driver uses for
 * sequences that arrived over xGMI -- fastcar's outer loop hands each chunk of queries to every database chunk
 * (fastcar/FC_Runner.cpp:585-597); with the database sharded over GPUs the chunk's bases are all-gathered device to device and
 * each rank builds the block's histograms itself. `packed` is read on the context's own stream: its bytes must be COMPLETE when the
 * call is made (a producer on another stream -- an RCCL all-gather -- is waited for by the caller first). */
int msc_hist_build_packed_dev(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs,
                              const void* packed_dev, uint64_t n_bases,
                              const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_segs,
                              const uint64_t* eff_len, const uint64_t* one_mers);

/* Debug / parity: bins in NATURAL k-mer order (first base most significant), 4^k * dtype/8 bytes. */
int msc_hist_download(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, void* bins_out);
int msc_hist_upload(msc_ctx* ctx, msc_hist_set* set, uint64_t slot, const void* bins, uint64_t length,
                    const uint64_t* one_mers /* 4 or NULL */);       /* DivergencePoint(pts, len) ctor: mag recomputed */
int msc_hist_info_get(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, msc_hist_info* out);
/* DivergencePoint::get_length() of n consecutive slots at once (the driver's sort by length, cluster/CRunner.cpp:529-560, reads every
 * point's): served from the lengths the library remembers on the host, or one strided copy. */
int msc_hist_lengths(msc_ctx* ctx, const msc_hist_set* set, uint64_t first_slot, uint64_t n, uint64_t* lengths_out);
int msc_hist_set_id(msc_ctx* ctx, msc_hist_set* set, uint64_t slot, uint64_t id);

/* DivergencePoint::clone (clutil/DivergencePoint.h:35-43): full copy, mag re-summed. Used by Center (cluster/Center.h:13-40). */
int msc_hist_clone(msc_ctx* ctx, msc_hist_set* dst, uint64_t dst_slot, const msc_hist_set* src, uint64_t src_slot);
/* DivergencePoint::set (clutil/DivergencePoint.cpp:182-190; caller cluster/ClusterFactory.cpp:328,331):
 * bins, length and id are copied, `mag` is NOT. */
int msc_hist_assign(msc_ctx* ctx, msc_hist_set* dst, uint64_t dst_slot, const msc_hist_set* src, uint64_t src_slot);
/* center->set(*next) for many centres in one launch (the tail of every mean_shift_update of a round, cluster/ClusterFactory.cpp:328,331):
 * msc_hist_assign(dst, dst_slots[i], src, src_slots[i]) for i < n; the destination slots must be distinct. */
int msc_hist_assign_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n);

/* Exact slot copy: bins and EVERY scalar (incl. a stale mag). What a host container of Center objects needs when it
 * relocates them without going through clone() (std::vector growth of the device-side centre store). */
int msc_hist_copy(msc_ctx* ctx, msc_hist_set* dst, uint64_t dst_slot, const msc_hist_set* src, uint64_t src_slot);
/* The same for n slots in one launch per region (relocating or compacting a whole centre store: the vector<Center*> of
 * cluster/ClusterFactory.cpp:560-575 as it grows); the destination slots must be distinct. A sparse destination appends all the
 * lists or none (MSC_ERR_OOM). */
int msc_hist_copy_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n);
/* msc_hist_clone for n slots in one launch per region: the Center(c->clone()) of every cluster the accumulate stage opened
 * (cluster/ClusterFactory.cpp:560-575, cluster/Center.h:13-40) -- a centre is not read before the update stage, so a driver may
 * queue its clones and issue them together. */
int msc_hist_clone_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n);

/* ------------------------------------------------------------------ a5/a7: model (Feature<T> + GLM weights) */
/* Mirrors the state Predictor::read_from builds (predict/Predictor.cpp:125-185): combos are replayed through
 * Feature::add_feature (predict/Feature.cpp:102-128) so single-feature order == order of first appearance.
 *   combo_kind[c] in MSC_COMBO_*, combo_flags[c] = OR of its 1-2 single bits, weights[0] = intercept.
 *   single_flags/mins/maxs: the n_singles lines of the file (any order).
 *   bias: Predictor::set_bias global (predict/Predictor.cpp:307-313), default 0. */
int  msc_model_create(msc_ctx* ctx, int k, int n_combos, const int* combo_kind, const uint64_t* combo_flags,
                      const double* weights, int n_singles, const uint64_t* single_flags, const double* mins,
                      const double* maxs, double bias, msc_model** out);
/* Reads a `--dump` / weights.txt file (predict/Predictor.cpp:28-44,47-121). block 0 = classification, 1 = regression. */
int  msc_model_load(msc_ctx* ctx, const char* path, int block, msc_model** out);
int  msc_model_parse(msc_ctx* ctx, const char* text, int block, msc_model** out);
void msc_model_destroy(msc_model* m);
int  msc_model_k(const msc_model* m);
int  msc_model_n_singles(const msc_model* m);
int  msc_model_n_combos(const msc_model* m);
int  msc_model_single_flags(const msc_model* m, uint64_t* flags_out /* n_singles */);
void msc_model_set_bias(msc_model* m, double bias);

/* ------------------------------------------------------------------ a5/a6: pair scoring, 1 query x m candidates */
/* Raw statistics (predict/Feature.cpp, the 11 functions of SURVEY 8a-a6) of (cands[cand_slots[i]], q) for every
 * bit in feat_mask. raw_out is m x popcount(feat_mask), row i = candidate i, columns in ascending bit order.
 * Replaces Feature<T>::<raw fn> call sites predict/Feature.cpp:156-171 and FeatureSelector's table
 * (predict/FeatureSelector.cpp:23-33). cand_slots == NULL means slots 0..m-1. */
int msc_pair_features_raw(msc_ctx* ctx, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                          const msc_hist_set* qset, uint64_t q_slot, int order, uint64_t feat_mask, double* raw_out);

/* Feature::compute + operator() + weighted sum (predict/Feature.h:197-239; cluster/Trainer.cpp:112-120;
 * predict/Predictor.cpp:284-333). Any output pointer may be NULL.
 *   singles_out : m x n_singles normalised values (the `cache` vector)
 *   combos_out  : m x n_combos
 *   sum_out     : m, s = w0 + sum_c w_c * combo_c
 *   csum_out    : m, classify_sum = logistic(s) + bias
 * Rows of candidates for which the reference would throw hold NaN and the call returns MSC_ERR_ZERO_LENGTH/NAN. */
int msc_score(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
              const msc_hist_set* qset, uint64_t q_slot, int order,
              double* singles_out, double* combos_out, double* sum_out, double* csum_out);

/* n_q queries x m candidates in ONE pass over the candidates (the all-pairs shape: fastcar work(),
 * fastcar/FC_Runner.cpp:426-471; the training table, predict/FeatureSelector.cpp:23-33). Every candidate tile read from
 * HBM is scored against several query tiles held in registers. Outputs are query-major: [n_q][m] (raw_out
 * [n_q][m][popcount(feat_mask)]). Any output pointer may be NULL; model may be NULL when only raw_out is wanted.
 * close_out[q][i] = round(classify_sum) > 0. */
int msc_score_multi(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                    const msc_hist_set* qset, const uint32_t* q_slots, uint64_t n_q, int order,
                    double* sum_out, double* csum_out, uint8_t* close_out, uint64_t feat_mask, double* raw_out);
/* counts[q] = number of close candidates of query q in the last msc_score_multi call that asked for close_out (the row sums of
 * close_out, added up on the device: the per-query result of fastcar's work() loop, fastcar/FC_Runner.cpp:449-455, without a host pass
 * over n_q x m flags). MSC_ERR_UNSUPPORTED when that call took a route that keeps no counts (one pass per query). */
int msc_last_close_counts(msc_ctx* ctx, uint64_t* counts, uint64_t n_q);

/* ------------------------------------------------------------------ a8/a9: Trainer operators */
/* Trainer<T>::get_close (cluster/Trainer.cpp:23-71; caller cluster/ClusterFactory.cpp:566).
 * The window [istart, iend) is cand_slots[0..m). Candidates outside floor(len_q*cutoff) <= len <= floor(len_q/cutoff)
 * are skipped. close_flags[i] = 1 where the reference sets (*i).second = true. best_pos = index INTO cand_slots of the
 * arg-max of combo 0 (first maximum in window order = OMP_NUM_THREADS=1 order), -1 if nothing passed the length
 * filter (then best_sim = -1). is_min = no candidate was close. `cutoff` is Trainer::cutoff as given (not get_id()). */
int msc_get_close(msc_ctx* ctx, const msc_model* model, double cutoff,
                  const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                  const msc_hist_set* qset, uint64_t q_slot,
                  uint8_t* close_flags, int64_t* best_pos, double* best_sim, int* is_min);

/* Trainer<T>::filter (cluster/Trainer.cpp:123-141; caller cluster/ClusterFactory.cpp:312): keep[i] = 1 iff point i
 * is inside the length window of the centre (get_id() form of cutoff) AND round(classify(centre, point)) != 0. */
int msc_filter(msc_ctx* ctx, const msc_model* model, double cutoff,
               const msc_hist_set* centre_set, uint64_t centre_slot,
               const msc_hist_set* pts, const uint32_t* pt_slots, uint64_t m, uint8_t* keep, uint64_t* n_kept);

/* Trainer<T>::merge (cluster/Trainer.cpp:74-109; caller cluster/ClusterFactory.cpp:387): among centres
 * centre_slots[begin..last] inside the length window of centre_slots[current] with round(classify_sum) == 1,
 * the index with the largest combo 0 (later index wins ties; initial best (0, DBL_MIN)). */
int msc_merge(msc_ctx* ctx, const msc_model* model, double cutoff,
              const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
              int64_t current, int64_t begin, int64_t last, int64_t* best_out);

/* Predictor::close + similarity for one query against m database entries (fastcar/FC_Runner.cpp:426-471 work()):
 * close_out[i] = p_close (classification block), sim_out[i] = p_predict (regression block, clamped to [0,1]).
 * work() follows Predictor::get_mode() (:432,446-458): cls == NULL (a `mode: 2` weights file) makes every entry close,
 * reg == NULL (a `mode: 1` file, what `meshclust2 --dump` writes) makes every similarity 1. Not both NULL. */
int msc_search(msc_ctx* ctx, const msc_model* cls, const msc_model* reg,
               const msc_hist_set* db, const uint32_t* db_slots, uint64_t m,
               const msc_hist_set* qset, uint64_t q_slot, uint8_t* close_out, double* sim_out);

/* ------------------------------------------------------------------ a8 over a device-resident window
 * The accumulate loop (cluster/ClusterFactory.cpp:553-610) hands Trainer::get_close an iterator range of the length-sorted store
 * (bvec::get_range, cluster/bvec.cpp:261-330; the loop `for (i = istart; i < iend; ++i)` of cluster/Trainer.cpp:41-48). A
 * msc_window keeps that store's ORDER on the device: position i holds slot slots[i] of `set` and an alive flag, all alive at first.
 * A step then passes the positions [first, end) instead of a slot list rebuilt on the host: host work per step is O(close).
 *   msc_get_close_window: get_close over the ALIVE positions of [first, end) in position order; same window rule, arg-max and
 *     tie order as msc_get_close. *close_pos (valid until the next call on this window) lists the n_close positions the reference
 *     would mark, ascending; they DIE with the call -- the loop removes marked points next (remove_available,
 *     cluster/ClusterFactory.cpp:598). best_pos = POSITION of the arg-max, -1 if nothing passed the length filter.
 *   msc_window_kill: positions that leave the store otherwise (bvec::erase of the next seed :589, bvec::pop :593).
 *   msc_window_alive: alive positions in [first, end), from the host-side count the library keeps (no device read-back). */
typedef struct msc_window msc_window;
int      msc_window_create(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* slots, uint64_t n, msc_window** out);
void     msc_window_destroy(msc_window* w);
uint64_t msc_window_alive(const msc_window* w, uint64_t first, uint64_t end);
int      msc_window_kill(msc_ctx* ctx, msc_window* w, const uint32_t* positions, uint64_t n);
int      msc_get_close_window(msc_ctx* ctx, const msc_model* model, double cutoff, msc_window* w, uint64_t first, uint64_t end,
                              const msc_hist_set* qset, uint64_t q_slot, const uint32_t** close_pos, uint64_t* n_close,
                              int64_t* best_pos, double* best_sim, int* is_min);

/* ------------------------------------------------------------------ a4/a10: mean-shift metric */
/* get_mean (cluster/ClusterFactory.cpp:338-380), the mean part of mean_shift_update (:297-326) and Trainer::closest
 * (cluster/Trainer.cpp:144-157): FP64 column mean of the m members, DivergencePoint::distance_d
 * (clutil/DivergencePoint.cpp:55-66) of every member to it, first arg-min.
 * dist_out (m) and mean_out (4^k doubles, natural order) may be NULL. */
int msc_mean_nearest(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m,
                     int64_t* nearest_pos, double* dist_out, double* mean_out);

/* The whole update stage of one mean-shift round in three launches: for centre c (slot centre_slots[c] of `centres`) and its
 * neighbourhood list pt_slots[offsets[c] .. offsets[c+1]) of `pts`: Trainer::filter, the FP64 mean of the survivors and
 * the survivor nearest that mean -- mean_shift_update, cluster/ClusterFactory.cpp:288-335, which the reference runs for all
 * centres of a round under `omp parallel for` (:639). nearest_pos[c] = position INSIDE centre c's list of the new centre
 * point, or -1 when nothing survives the filter; n_kept[c] (nullable) = survivors. Same results as msc_filter followed by
 * msc_mean_nearest per centre (which is also what runs for divergence / group statistics, the wide range, and sparse sets with
 * counts >= 2^16; sparse sets in the 32-bit range are batched too since r02). */
int msc_update_centres(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                       uint64_t n_centres, const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, int64_t* nearest_pos,
                       uint64_t* n_kept);

/* Every Trainer::merge call of the serial merge loop (cluster/ClusterFactory.cpp:383-401: centre i against centres i+1 ..
 * min(n-1, i+delta), i = 0 .. n-1) in one launch; best_out[i] = what msc_merge returns for (current = i, begin = i + 1,
 * last = min(n - 1, i + delta)). No merge call changes a histogram, so the calls are independent. */
int msc_merge_all(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
                  int delta, int64_t* best_out);

/* ... for SOME of those calls: best_out[w] = what msc_merge returns for (current = which[w], begin = which[w] + 1,
 * last = min(n - 1, which[w] + delta)) over the same list of n centres. Once the clusters have settled a round of the merge loop
 * repeats the round before it; the driver asks only about the centres whose delta + 1 histograms changed. */
int msc_merge_some(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
                   int delta, const uint64_t* which, uint64_t n_which, int64_t* best_out);

/* ------------------------------------------------------------------ f2: training on labelled pairs
 * Replaces the feature-selection half of Predictor<T>::train (predict/Predictor.cpp:876-975): calculate_table,
 * BestFirstSelector<T>::train_class and GLM::train (predict/BestFirstSelector.cpp:113-250, predict/GLM.cpp:20-23) on pairs
 * (first_slots[i], second_slots[i]) of `pts` labelled vals[i] (identity in the unit of `id`); the first n_train pairs are the
 * training set, the next n_test the testing set. feat_flags = the single statistics on offer (MSC_FEAT_FAST / MSC_FEAT_SLOW).
 * The raw statistics come from the streaming kernels; selection and least squares are host work in the reference's evaluation
 * order. text_out receives a complete weights file (Predictor::save, :28-44) that msc_model_parse accepts. The reference's own
 * generation of the pairs (mutated templates, :519-710) is out of scope. */
int msc_train_class(msc_ctx* ctx, const msc_hist_set* pts, const uint32_t* first_slots, const uint32_t* second_slots, const double* vals,
                    uint64_t n_train, uint64_t n_test, uint64_t feat_flags, int min_feat, int max_feat, double id, char* text_out, size_t cap,
                    double* train_acc, double* test_acc);

/* The regression half of Predictor<T>::train (predict/Predictor.cpp:977-985 train_regr -> GreedySelector<T>::train_regression,
 * predict/GreedySelector.cpp:11-76): the model fastcar's work() reads identities from (Predictor::similarity -> p_predict, :284-300).
 * Same inputs as msc_train_class; vals are the identity labels the least-squares fit predicts. Greedy forward selection over the
 * candidates of Predictor::add_feats: at most max_feat rounds, a round keeps the candidate with the smallest mean absolute error over the
 * TESTING pairs if that beats every earlier round. text_out: a complete weights file with `mode: 2` (msc_model_parse(text, 1) reads
 * the block); train_err / test_err: mean |prediction - label| of the final fit. The reference's own function cannot run to its end
 * (no return statement: `fastcar --dump` dies there); oracle/ref_harness.cpp follows its body on the reference's objects and
 * tests/golden/train_regr_*.json holds the result. */
int msc_train_regr(msc_ctx* ctx, const msc_hist_set* pts, const uint32_t* first_slots, const uint32_t* second_slots, const double* vals,
                   uint64_t n_train, uint64_t n_test, uint64_t feat_flags, int max_feat, double id, char* text_out, size_t cap,
                   double* train_err, double* test_err);

/* ------------------------------------------------------------------ multi-GPU plumbing
 * Raw device views so that a caller that owns an RCCL communicator (torch.distributed / rccl.h) can broadcast a
 * query or all-gather centroid histograms between the per-GPU processes without a host bounce.
 *   slot_bytes  : bytes of one slot's bins region; slot i starts at bins + i*slot_bytes
 *   scalar_bytes: bytes of one slot's scalar record; slot i at scalars + i*scalar_bytes
 * After writing a slot's two regions from a peer's copy, call msc_hist_import_done() for those slots. */
int msc_hist_set_device_view(const msc_hist_set* set, void** bins, uint64_t* slot_bytes,
                             void** scalars, uint64_t* scalar_bytes);
int msc_hist_import_done(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n);

/* A slot as ONE contiguous byte range in device memory, dense or sparse (scalar record + bins, or scalar record + sub-range table +
 * (bin, value) list + cum array: ~12 bytes per distinct k-mer instead of 4^k bins): what a sharded driver broadcasts as the query of
 * a Trainer::get_close step (cluster/ClusterFactory.cpp:566) and all-gathers as the new centres of an update round (:328,331).
 *   msc_hist_packed_bytes: size of slot's range (a multiple of 16; known on the host).
 *   msc_hist_pack:   slots[i] -> dev_dst + offsets[i] (offsets: multiples of 16, ranges disjoint).
 *   msc_hist_unpack: dev_src + offsets[i] -> slots[i] of `set` (same k, bin type and layout), an exact copy -- stale magnitude
 *                    included; clone / assign semantics come from msc_hist_clone / msc_hist_assign(_batch) out of a staging set. A
 *                    sparse destination appends the lists to its arena (MSC_ERR_OOM when full).
 *   msc_hist_set_reset: a sparse set forgets every list (its arena is append-only otherwise): what a one-slot staging set does
 *                    between two queries. */
uint64_t msc_hist_packed_bytes(const msc_hist_set* set, uint64_t slot);
int      msc_hist_pack(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* slots, uint64_t n, void* dev_dst, const uint64_t* offsets);
int      msc_hist_unpack(msc_ctx* ctx, msc_hist_set* set, const uint32_t* slots, uint64_t n, const void* dev_src, const uint64_t* offsets);
int      msc_hist_set_reset(msc_ctx* ctx, msc_hist_set* set);

/* get_mean / the mean of mean_shift_update (cluster/ClusterFactory.cpp:338-380,297-326) when the members of a list live on several
 * GPUs: SURVEY 8(e)'s reduction of partial column sums. n lists at once; list c = member_slots[offsets[c] .. offsets[c+1]) of THIS
 * rank's set (may be empty).
 *   msc_colsum_partial: the integer column sums of this rank's members. *dev_payload (library-owned, valid until the next call) is
 *     dense sets : uint64 [n][padded bins] followed by uint64 [n] member counts -> ALL-REDUCE (sum) it in place, payload_bytes / 8 words;
 *     sparse sets: a table {n, bytes, (members, offset) x n} and per list a packed sparse slot of the summed excesses -> ALL-GATHER
 *                  it, every rank padded to the largest payload_bytes.
 *   msc_colsum_nearest: dev_global = the reduced array (dense; world ignored) or the gathered payloads, bytes_per_rank apart (sparse).
 *     Every rank derives the same FP64 mean of each list (exact integer sums / total members: the single-GPU mean bit for bit),
 *     rounds it, and measures ITS members: nearest_pos[c] = position inside this rank's list c of its first member nearest the mean
 *     (DivergencePoint::distance_d, clutil/DivergencePoint.cpp:55-66), -1 for an empty list; nearest_dist[c] its distance;
 *     m_total_out[c] (nullable) = members of list c over all ranks. The caller folds the ranks' (distance, position) records.
 *   msc_colsum_list_bytes: device scratch one list costs (a driver sizes its chunks of centres by it).
 * msc_filter_batch: Trainer::filter (cluster/Trainer.cpp:123-141) of n centres against their lists in one pass -- the first stage of
 *   msc_update_centres alone; keep[i] = 1 iff pt_slots[i] survives the filter of its centre. */
uint64_t msc_colsum_list_bytes(const msc_hist_set* set);
int      msc_colsum_partial(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, const uint64_t* offsets, uint64_t n,
                            void** dev_payload, uint64_t* payload_bytes);
int      msc_colsum_nearest(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, const uint64_t* offsets, uint64_t n,
                            const void* dev_global, uint64_t bytes_per_rank, int world, int64_t* nearest_pos, double* nearest_dist,
                            uint64_t* m_total_out);
int      msc_filter_batch(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                          uint64_t n_centres, const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, uint8_t* keep);

/* Plain device memory and the ctx's HIP stream for a caller that owns the communicator (rccl.h takes a hipStream_t: collectives
 * queued on this stream are ordered with the library's kernels without a host round trip). */
void*    msc_stream_handle(msc_ctx* ctx);
int      msc_device_malloc(msc_ctx* ctx, uint64_t bytes, void** out);
int      msc_device_free(msc_ctx* ctx, void* p);
int      msc_memcpy_to_host(msc_ctx* ctx, void* dst, const void* src_dev, uint64_t bytes);
int      msc_memcpy_to_device(msc_ctx* ctx, void* dst_dev, const void* src, uint64_t bytes);
int      msc_memcpy_device(msc_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes);      /* queued on the ctx stream, not waited for */
/* Page-locked host memory for result arrays that are filled call after call (the [n_q][m] outputs of msc_score_multi: into pageable
 * memory the runtime stages every copy through its own bounce buffers -- 0.6 ms of a 7 ms step for 6.4 MB of close flags). */
int      msc_host_alloc(msc_ctx* ctx, uint64_t bytes, void** out);
int      msc_host_free(msc_ctx* ctx, void* p);

#ifdef __cplusplus
}
#endif
#endif /* MESHCLUST2_HIP_H */
