/* oracle/msc_oracle.h -- TEST INFRASTRUCTURE ONLY. NOT part of the product.
 *
 * Plain-C CPU restatement of MeShClust2's alignment-free pairwise-identity hot path
 * (SURVEY.md section 8a rows a1-a10). Every function cites the reference file:line it follows.
 * It mirrors the reference's PASS STRUCTURE on purpose (one full pass per raw feature, compute()
 * twice per get_close pair) so that it can double as the "port" CPU baseline in bench.py.
 *
 * Parity status: PINNED. tests/test_oracle_vs_ref.py compares every function below with the real
 * reference compiled from /root/reference (oracle/_ref/libmsc_ref.so, oracle/ref_harness.cpp) on
 * seeded inputs, and tests/test_oracle_golden.py checks it against the committed fixtures in
 * tests/golden/ that were generated from that same reference build (tests/golden/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef MSC_ORACLE_H
#define MSC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Single-feature flag bits, predict/Feature.h:31-64 (only the 11 in scope). */
#define ORC_FEAT_MANHATTAN          (1ULL << 2)
#define ORC_FEAT_EUCLIDEAN          (1ULL << 3)
#define ORC_FEAT_NORMALIZED_VECTORS (1ULL << 5)
#define ORC_FEAT_JEFFEREY_DIV       (1ULL << 7)
#define ORC_FEAT_PEARSON_COEFF      (1ULL << 9)
#define ORC_FEAT_INTERSECTION       (1ULL << 13)
#define ORC_FEAT_EMD                (1ULL << 18)
#define ORC_FEAT_LENGTHD            (1ULL << 21)
#define ORC_FEAT_KULCZYNSKI2        (1ULL << 27)
#define ORC_FEAT_SIMRATIO           (1ULL << 28)
#define ORC_FEAT_JENSEN_SHANNON     (1ULL << 29)
/* two of the reference's `extraslow` statistics (PRED_FEAT_ALL, predict/Predictor.h:25) that BASELINE's north_star names */
#define ORC_FEAT_RRE_K_R            (1ULL << 14)
#define ORC_FEAT_SIM_MM             (1ULL << 16)
/* predict/Predictor.h:23-24 */
#define ORC_FEAT_FAST (ORC_FEAT_EUCLIDEAN | ORC_FEAT_MANHATTAN | ORC_FEAT_INTERSECTION | ORC_FEAT_KULCZYNSKI2 | \
                       ORC_FEAT_SIMRATIO | ORC_FEAT_NORMALIZED_VECTORS | ORC_FEAT_PEARSON_COEFF | ORC_FEAT_EMD | ORC_FEAT_LENGTHD)
#define ORC_FEAT_DIV  (ORC_FEAT_JEFFEREY_DIV | ORC_FEAT_JENSEN_SHANNON)

#define ORC_MAX_SINGLES 34
#define ORC_MAX_COMBOS  16

/* One k-mer histogram = the data members of DivergencePoint<T> (clutil/DivergencePoint.h:14-88)
 * that the path reads. bins are in natural k-mer order (first base most significant). */
typedef struct {
	int       dtype;        /* 8, 16, 32, 64 */
	int       k;
	uint64_t  nbins;        /* 4^k */
	void*     bins;         /* nbins * dtype/8 bytes, owned */
	uint64_t  mag;          /* DivergencePoint::mag, set ONLY by the ctor (stale after set()) */
	uint64_t  length;       /* effective length (sum of segment lengths) */
	double    stddev;
	uint64_t  one_mers[4];  /* k=1 table, pseudocount 1 */
	int       overflow;     /* 1 if any bin saturated at max(T) */
	uint64_t  id;
} orc_hist;

/* Feature<T> state + GLM weights of one block of a weights file (predict/Predictor.cpp:82-185). */
typedef struct {
	int      k;
	int      n_singles;
	uint64_t single_flag[ORC_MAX_SINGLES];
	double   mins[ORC_MAX_SINGLES], maxs[ORC_MAX_SINGLES];
	int      is_sim[ORC_MAX_SINGLES];
	int      n_combos;
	int      combo_kind[ORC_MAX_COMBOS];      /* FILE codes: 0 xy, 1 xy2, 2 x2y, 3 x2y2 */
	int      combo_n[ORC_MAX_COMBOS];
	int      combo_idx[ORC_MAX_COMBOS][ORC_MAX_SINGLES];
	uint64_t combo_flags[ORC_MAX_COMBOS];
	double   weights[ORC_MAX_COMBOS + 1];     /* [0] = intercept */
} orc_model;

typedef struct {
	int       k;
	int       mode;          /* bit0 classification block, bit1 regression block */
	int       max_features;
	double    id;
	char      datatype[16];
	uint64_t  feature_set;
	orc_model cls, reg;
	double    bias;          /* predict/Predictor.cpp:307 global _bias, default 0 */
} orc_predictor;

/* ---- a1: sequence encoding (nonltr/Chromosome.cpp:130-154,263-385; ChromosomeOneDigit.cpp:79-133) ---- */
/* codes_out: same length as seq; 0..3 inside/near segments, raw 'N' kept where the reference keeps it.
 * segs_out: [s0,e0,s1,e1,...] inclusive. Returns number of segments, -1 on invalid character. */
int  orc_encode(const char* seq, size_t len, char* codes_out, int64_t* segs_out, size_t max_segs, uint64_t* eff_len);

/* ---- a2+a3: histogram build (nonltr/KmerHashTable.cpp:33-84,134-160,236-256; clutil/Loader.cpp:42-86,112-179) ---- */
int  orc_hist_build(const char* seq, size_t len, int k, int dtype, int strip_non_acgt, orc_hist* out);
void orc_hist_free(orc_hist* h);
int  orc_hist_clone(const orc_hist* src, orc_hist* dst);      /* DivergencePoint::clone, DivergencePoint.h:35-43 (recomputes mag) */
void orc_hist_set(orc_hist* dst, const orc_hist* src);        /* DivergencePoint::set, DivergencePoint.cpp:182-190 (mag NOT copied) */

/* ---- a6: the 11 raw statistics, argument order (a, b) as at the call site ---- */
/* returns NaN for an unknown flag; sets *err=123 for length_difference with a zero length (Feature.cpp:878-886) */
double orc_raw_feature(uint64_t flag, const orc_hist* a, const orc_hist* b, int* err);

/* ---- a5+a7: compute / normalise / combos / GLM ---- */
int    orc_predictor_parse(const char* text, orc_predictor* out);              /* Predictor.cpp:47-79,125-185 */
int    orc_predictor_load(const char* path, orc_predictor* out);
int    orc_predictor_format(const orc_predictor* p, char* buf, size_t cap);    /* Predictor.cpp:28-44,82-121 */
int    orc_model_add_feature(orc_model* m, uint64_t flags, int file_combo_code); /* Feature.cpp:102-128 */
int    orc_model_set_normal(orc_model* m, uint64_t flag, double mn, double mx);  /* Feature.cpp:173-180 */
int    orc_feat_is_sim(uint64_t flag);                                           /* Feature.cpp:549-663 */
/* Feature::compute (Feature.h:197-201): singles[n_singles] normalised; returns 0, or -1 on NaN (Feature.cpp:143-146) */
int    orc_compute(const orc_model* m, const orc_hist* a, const orc_hist* b, double* singles);
double orc_combo(const orc_model* m, int col, const double* singles);            /* Feature.h:205-239 */
double orc_weighted_sum(const orc_model* m, const double* singles);              /* Trainer.cpp:112-119 */
double orc_logistic(double x);                                                   /* GLM.cpp:26-29 */
double orc_classify(const orc_predictor* p, const orc_hist* a, const orc_hist* b); /* Trainer.cpp:112-120 -> Predictor.cpp:315-320 */
int    orc_p_close(const orc_predictor* p, const orc_hist* a, const orc_hist* b);  /* Predictor.cpp:323-333 */
double orc_p_predict(const orc_predictor* p, const orc_hist* a, const orc_hist* b);/* Predictor.cpp:284-300 */

/* ---- a8/a9: Trainer operators (cluster/Trainer.cpp) ---- */
/* get_close :23-71. cands[0..m) is the window [istart, iend). flags[i]=1 where the reference sets (*i).second=true.
 * best_pos=-1 / best_sim=-1 when nothing passed the length filter. Serial (OMP_NUM_THREADS=1) tie order. */
int  orc_get_close(const orc_predictor* p, double cutoff, const orc_hist* query, const orc_hist* const* cands, size_t m,
                   uint8_t* flags, int64_t* best_pos, double* best_sim, int* is_min);
/* same result, OpenMP over candidates like the reference (Trainer.cpp:41); used for the CPU baseline timing */
int  orc_get_close_omp(const orc_predictor* p, double cutoff, const orc_hist* query, const orc_hist* const* cands, size_t m,
                       uint8_t* flags, int64_t* best_pos, double* best_sim, int* is_min);
/* filter :123-141: keep[i]=1 iff pts[i] survives; returns number kept */
int  orc_filter(const orc_predictor* p, double cutoff, const orc_hist* centre, const orc_hist* const* pts, size_t m, uint8_t* keep);
/* merge :74-109 */
long orc_merge(const orc_predictor* p, double cutoff, const orc_hist* const* centres, size_t n, long current, long begin, long last);

/* ---- a4/a10: mean-shift metric ---- */
double   orc_distance_d(const orc_hist* a, const double* mean);                  /* DivergencePoint.cpp:55-66 */
uint64_t orc_distance(const orc_hist* a, const orc_hist* b);                     /* DivergencePoint.cpp:70-82 */
/* get_mean (ClusterFactory.cpp:338-380) / closest (Trainer.cpp:144-157): mean[nbins], dists[m], first arg-min */
int  orc_mean_nearest(const orc_hist* const* pts, size_t m, double* mean_out, double* dists, int64_t* nearest);

void orc_set_threads(int n);
int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
