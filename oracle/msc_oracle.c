/* oracle/msc_oracle.c -- TEST INFRASTRUCTURE ONLY. NOT part of the product; see msc_oracle.h.
 *
 * CPU restatement of the MeShClust2 pairwise-identity hot path, one function per reference
 * function, each citing the reference file:line (relative to /root/reference/src) it follows.
 * Compiled with -ffp-contract=off so FP64 expressions round exactly as written.
 */
#define _GNU_SOURCE
#include "msc_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ per-type bodies */
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)

#define T uint8_t
#define PROMO int
#define TMAX UINT8_MAX
#define FN(x) CAT(x, u8)
#define PROMO_MUL(a, b) ((int64_t)(int32_t)((uint32_t)(a) * (uint32_t)(b)))
#define PROMO_SQ(a, b) ((int64_t)(int32_t)((uint32_t)((int)(a) - (int)(b)) * (uint32_t)((int)(a) - (int)(b))))
#include "msc_oracle_t.inc"
#undef T
#undef PROMO
#undef TMAX
#undef FN
#undef PROMO_MUL
#undef PROMO_SQ

#define T uint16_t
#define PROMO int
#define TMAX UINT16_MAX
#define FN(x) CAT(x, u16)
#define PROMO_MUL(a, b) ((int64_t)(int32_t)((uint32_t)(a) * (uint32_t)(b)))
#define PROMO_SQ(a, b) ((int64_t)(int32_t)((uint32_t)((int)(a) - (int)(b)) * (uint32_t)((int)(a) - (int)(b))))
#include "msc_oracle_t.inc"
#undef T
#undef PROMO
#undef TMAX
#undef FN
#undef PROMO_MUL
#undef PROMO_SQ

#define T uint32_t
#define PROMO uint32_t
#define TMAX UINT32_MAX
#define FN(x) CAT(x, u32)
#define PROMO_MUL(a, b) ((uint32_t)((uint32_t)(a) * (uint32_t)(b)))
#define PROMO_SQ(a, b) ((uint32_t)((uint32_t)((a) - (b)) * (uint32_t)((a) - (b))))
#include "msc_oracle_t.inc"
#undef T
#undef PROMO
#undef TMAX
#undef FN
#undef PROMO_MUL
#undef PROMO_SQ

#define T uint64_t
#define PROMO uint64_t
#define TMAX UINT64_MAX
#define FN(x) CAT(x, u64)
#define PROMO_MUL(a, b) ((uint64_t)((uint64_t)(a) * (uint64_t)(b)))
#define PROMO_SQ(a, b) ((uint64_t)((uint64_t)((a) - (b)) * (uint64_t)((a) - (b))))
#include "msc_oracle_t.inc"
#undef T
#undef PROMO
#undef TMAX
#undef FN
#undef PROMO_MUL
#undef PROMO_SQ

#define BY_TYPE(dt, call8, call16, call32, call64) \
	((dt) == 8 ? (call8) : (dt) == 16 ? (call16) : (dt) == 32 ? (call32) : (call64))

/* ------------------------------------------------------------------ a1: encoding */

/* ChromosomeOneDigitDna::buildCodes, nonltr/ChromosomeOneDigitDna.cpp:48-68 */
static int dna_code(char c) {
	switch (c) {
	case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
	case 'R': return 2; case 'Y': return 1; case 'M': return 0; case 'K': return 3;
	case 'S': return 2; case 'W': return 3; case 'H': return 1; case 'B': return 3;
	case 'V': return 0; case 'D': return 3; case 'N': return 1; case 'X': return 2;
	default: return -1;
	}
}

typedef struct { int64_t s, e; } seg_t;

int orc_encode(const char* seq, size_t len, char* base, int64_t* segs_out, size_t max_segs, uint64_t* eff_len) {
	const int64_t n = (int64_t)len;
	const int seg_length = 1000000;                         /* Chromosome::finalize -> help(1000000, true), Chromosome.cpp:115-128 */
	seg_t* seg = (seg_t*)malloc(sizeof(seg_t) * (len / 2 + 2));
	size_t nseg = 0;
	if (!seg) return -1;

	/* toUpperCase, Chromosome.cpp:252-256 */
	for (int64_t i = 0; i < n; i++) base[i] = (char)toupper((unsigned char)seq[i]);

	/* removeAmbiguous, Chromosome.cpp:263-291 (if / else-if chain kept: a run that STARTS on the
	 * last character is never closed) */
	int64_t start = -1;
	for (int64_t i = 0; i < n; i++) {
		if (base[i] != 'N' && start == -1) {
			start = i;
		} else if (base[i] == 'N' && start != -1) {
			seg[nseg].s = start; seg[nseg].e = i - 1; nseg++;
			start = -1;
		} else if (i == n - 1 && base[i] != 'N' && start != -1) {
			seg[nseg].s = start; seg[nseg].e = i; nseg++;
			start = -1;
		}
	}

	/* mergeSegments, Chromosome.cpp:298-353, only if base.size() > 20 (:146) */
	if (n > 20 && nseg > 0) {
		seg_t* m = (seg_t*)malloc(sizeof(seg_t) * (nseg + 1));
		size_t nm = 0;
		int64_t s = seg[0].s, e = seg[0].e;
		for (size_t i = 1; i < nseg; i++) {
			int64_t s1 = seg[i].s, e1 = seg[i].e;
			if (s1 - e < 10) {
				e = e1;
			} else {
				if (e - s + 1 >= 20) { m[nm].s = s; m[nm].e = e; nm++; }
				s = s1; e = e1;
			}
		}
		if (e - s + 1 >= 20) { m[nm].s = s; m[nm].e = e; nm++; }
		free(seg);
		seg = m;
		nseg = nm;
	}

	/* makeSegmentList, Chromosome.cpp:355-385 */
	{
		size_t cap = nseg + (size_t)(n / seg_length) + 2;
		seg_t* l = (seg_t*)malloc(sizeof(seg_t) * cap);
		size_t nl = 0;
		for (size_t o = 0; o < nseg; o++) {
			int64_t s = seg[o].s, e = seg[o].e;
			if (e - s + 1 > seg_length) {
				int frag_num = (int)((e - s + 1) / seg_length);
				for (int h = 0; h < frag_num; h++) {
					int64_t fs = s + (int64_t)h * seg_length;
					int64_t fe = (h == frag_num - 1) ? e : fs + seg_length - 1;
					l[nl].s = fs; l[nl].e = fe; nl++;
				}
			} else {
				l[nl].s = s; l[nl].e = e; nl++;
			}
		}
		free(seg);
		seg = l;
		nseg = nl;
	}

	/* calculateEffectiveSize */
	uint64_t eff = 0;
	for (size_t i = 0; i < nseg; i++) eff += (uint64_t)(seg[i].e - seg[i].s + 1);
	if (eff_len) *eff_len = eff;

	/* ChromosomeOneDigit::encode, nonltr/ChromosomeOneDigit.cpp:79-133 */
	for (size_t s = 0; s < nseg; s++) {
		for (int64_t i = seg[s].s; i <= seg[s].e; i++) {
			int c = dna_code(base[i]);
			if (c < 0) { free(seg); return -1; }               /* InvalidInputException :86-94 */
			base[i] = (char)c;
		}
	}
	if (nseg > 0) {
		int64_t a = 0, b = seg[0].s - 1;
		for (size_t s = 0; s <= nseg; s++) {
			for (int64_t i = a; i <= b; i++) {
				char c = base[i];
				if (c != 'N') {
					int code = dna_code(c);
					if (code < 0) { free(seg); return -1; }    /* :110-117 */
					base[i] = (char)code;
				}
			}
			if (s + 1 < nseg) { a = seg[s].e + 1; b = seg[s + 1].s - 1; }
			else if (s + 1 == nseg) { a = seg[s].e + 1; b = n - 1; }
		}
	}

	for (size_t i = 0; i < nseg && i < max_segs; i++) { segs_out[2 * i] = seg[i].s; segs_out[2 * i + 1] = seg[i].e; }
	free(seg);
	return (int)nseg;
}

/* ------------------------------------------------------------------ a2/a3: histogram build */

static uint64_t pow4(int k) { uint64_t n = 1; for (int i = 0; i < k; i++) n *= 4; return n; }

int orc_hist_build(const char* seq_in, size_t len_in, int k, int dtype, int strip, orc_hist* out) {
	memset(out, 0, sizeof(*out));
	if (k < 1 || k > 15 || !(dtype == 8 || dtype == 16 || dtype == 32 || dtype == 64)) return -2;
	char* seq = (char*)malloc(len_in + 1);
	size_t len = 0;
	if (strip) {            /* Loader::get_point(string...), clutil/Loader.cpp:112-134: keep only upper-case ACGT */
		for (size_t i = 0; i < len_in; i++) {
			char c = seq_in[i];
			if (c == 'A' || c == 'C' || c == 'G' || c == 'T') seq[len++] = c;
		}
	} else {
		memcpy(seq, seq_in, len_in);
		len = len_in;
	}
	char* codes = (char*)malloc(len + 1);
	size_t max_segs = len / 2 + 2;
	int64_t* segs = (int64_t*)malloc(sizeof(int64_t) * 2 * max_segs);
	uint64_t eff = 0;
	int nseg = orc_encode(seq, len, codes, segs, max_segs, &eff);
	free(seq);
	if (nseg < 0) { free(codes); free(segs); return -1; }

	const uint64_t N = pow4(k);
	const size_t esz = (size_t)dtype / 8;
	void* bins = malloc(N * esz);
	/* KmerHashTable::initialize with initialValue 1, nonltr/KmerHashTable.cpp:69-72; Loader.cpp:141-144 */
	for (uint64_t i = 0; i < N; i++) {
		switch (dtype) {
		case 8: ((uint8_t*)bins)[i] = 1; break;
		case 16: ((uint16_t*)bins)[i] = 1; break;
		case 32: ((uint32_t*)bins)[i] = 1; break;
		default: ((uint64_t*)bins)[i] = 1; break;
		}
	}
	uint64_t k1[4] = {1, 1, 1, 1};
	int overflow = 0;
	/* Loader::fill_table, clutil/Loader.cpp:42-86 */
	for (int s = 0; s < nseg; s++) {
		int64_t a = segs[2 * s], b = segs[2 * s + 1];
		if (b - a + 1 >= k) {
			int r = BY_TYPE(dtype, count_segment_u8((uint8_t*)bins, codes, a, b - k + 1, k),
			                count_segment_u16((uint16_t*)bins, codes, a, b - k + 1, k),
			                count_segment_u32((uint32_t*)bins, codes, a, b - k + 1, k),
			                count_segment_u64((uint64_t*)bins, codes, a, b - k + 1, k));
			if (r == -1) overflow = 1;
		}
		if (b - a + 1 >= 1) count_segment_u64(k1, codes, a, b, 1);
	}
	free(codes);
	free(segs);

	out->dtype = dtype;
	out->k = k;
	out->nbins = N;
	out->bins = bins;
	out->length = eff;                                   /* p->set_length(chrom->getEffectiveSize()), Loader.cpp:156 */
	out->overflow = overflow;
	memcpy(out->one_mers, k1, sizeof(k1));
	switch (dtype) {
	case 8: finish_point_u8((uint8_t*)bins, N, &out->mag, &out->stddev); break;
	case 16: finish_point_u16((uint16_t*)bins, N, &out->mag, &out->stddev); break;
	case 32: finish_point_u32((uint32_t*)bins, N, &out->mag, &out->stddev); break;
	default: finish_point_u64((uint64_t*)bins, N, &out->mag, &out->stddev); break;
	}
	return 0;
}

void orc_hist_free(orc_hist* h) {
	if (h && h->bins) { free(h->bins); h->bins = NULL; }
}

int orc_hist_clone(const orc_hist* src, orc_hist* dst) {
	*dst = *src;
	size_t bytes = src->nbins * (size_t)src->dtype / 8;
	dst->bins = malloc(bytes);
	memcpy(dst->bins, src->bins, bytes);
	/* clone() goes through the (pts, len) ctor, which re-sums mag (DivergencePoint.h:35-43, .cpp:99-110) */
	double sd;
	switch (src->dtype) {
	case 8: finish_point_u8((uint8_t*)dst->bins, dst->nbins, &dst->mag, &sd); break;
	case 16: finish_point_u16((uint16_t*)dst->bins, dst->nbins, &dst->mag, &sd); break;
	case 32: finish_point_u32((uint32_t*)dst->bins, dst->nbins, &dst->mag, &sd); break;
	default: finish_point_u64((uint64_t*)dst->bins, dst->nbins, &dst->mag, &sd); break;
	}
	return 0;
}

void orc_hist_set(orc_hist* dst, const orc_hist* src) {
	/* DivergencePoint::set, clutil/DivergencePoint.cpp:182-190: points, length, id -- NOT mag, NOT stddev */
	memcpy(dst->bins, src->bins, src->nbins * (size_t)src->dtype / 8);
	dst->length = src->length;
	dst->id = src->id;
}

/* ------------------------------------------------------------------ a6: raw features */

double orc_raw_feature(uint64_t flag, const orc_hist* a, const orc_hist* b, int* err) {
	const size_t N = a->nbins;
	const int dt = a->dtype;
	const void *p = a->bins, *q = b->bins;
	if (err) *err = 0;
	switch (flag) {
	case ORC_FEAT_MANHATTAN:
		return BY_TYPE(dt, manhattan_u8(p, q, N), manhattan_u16(p, q, N), manhattan_u32(p, q, N), manhattan_u64(p, q, N));
	case ORC_FEAT_EUCLIDEAN:
		return BY_TYPE(dt, euclidean_u8(p, q, N), euclidean_u16(p, q, N), euclidean_u32(p, q, N), euclidean_u64(p, q, N));
	case ORC_FEAT_NORMALIZED_VECTORS:
		return BY_TYPE(dt, normalized_vectors_u8(p, q, N), normalized_vectors_u16(p, q, N), normalized_vectors_u32(p, q, N), normalized_vectors_u64(p, q, N));
	case ORC_FEAT_JEFFEREY_DIV:
		return BY_TYPE(dt, jefferey_u8(p, q, N, a->mag, b->mag), jefferey_u16(p, q, N, a->mag, b->mag), jefferey_u32(p, q, N, a->mag, b->mag), jefferey_u64(p, q, N, a->mag, b->mag));
	case ORC_FEAT_PEARSON_COEFF:
		return BY_TYPE(dt, pearson_u8(p, q, N, a->mag, b->mag), pearson_u16(p, q, N, a->mag, b->mag), pearson_u32(p, q, N, a->mag, b->mag), pearson_u64(p, q, N, a->mag, b->mag));
	case ORC_FEAT_INTERSECTION:
		return BY_TYPE(dt, intersection_u8(p, q, N, a->mag, b->mag), intersection_u16(p, q, N, a->mag, b->mag), intersection_u32(p, q, N, a->mag, b->mag), intersection_u64(p, q, N, a->mag, b->mag));
	case ORC_FEAT_EMD:
		return BY_TYPE(dt, emd_u8(p, q, N), emd_u16(p, q, N), emd_u32(p, q, N), emd_u64(p, q, N));
	case ORC_FEAT_LENGTHD: {
		/* Feature<T>::length_difference, predict/Feature.cpp:874-887 */
		uint64_t lp = a->length, lq = b->length;
		if (lp == 0 || lq == 0) { if (err) *err = 123; return NAN; }
		return (double)((lp > lq) ? (lp - lq) : (lq - lp));
	}
	case ORC_FEAT_KULCZYNSKI2:
		return BY_TYPE(dt, kulczynski2_u8(p, q, N, a->mag, b->mag), kulczynski2_u16(p, q, N, a->mag, b->mag), kulczynski2_u32(p, q, N, a->mag, b->mag), kulczynski2_u64(p, q, N, a->mag, b->mag));
	case ORC_FEAT_SIMRATIO:
		return BY_TYPE(dt, simratio_u8(p, q, N), simratio_u16(p, q, N), simratio_u32(p, q, N), simratio_u64(p, q, N));
	case ORC_FEAT_JENSEN_SHANNON:
		return BY_TYPE(dt, jensen_shannon_u8(p, q, N, a->mag, b->mag), jensen_shannon_u16(p, q, N, a->mag, b->mag), jensen_shannon_u32(p, q, N, a->mag, b->mag), jensen_shannon_u64(p, q, N, a->mag, b->mag));
	case ORC_FEAT_RRE_K_R:
		return BY_TYPE(dt, rre_k_r_u8(p, q, N), rre_k_r_u16(p, q, N), rre_k_r_u32(p, q, N), rre_k_r_u64(p, q, N));
	case ORC_FEAT_SIM_MM: {
		/* Feature<T>::sim_mm + d_markov, predict/Feature.cpp:1429-1455: d_markov(a, b) = log(markov(b, a) / markov(b, b)) /
		 * b.getRealMagnitude() (= mag - N, clutil/DivergencePoint.cpp:272-275); sim_mm = 1 - exp((d(a, b) + d(b, a)) / 2) */
		double m_ba = BY_TYPE(dt, markov_u8(q, p, N), markov_u16(q, p, N), markov_u32(q, p, N), markov_u64(q, p, N));
		double m_bb = BY_TYPE(dt, markov_u8(q, q, N), markov_u16(q, q, N), markov_u32(q, q, N), markov_u64(q, q, N));
		double m_ab = BY_TYPE(dt, markov_u8(p, q, N), markov_u16(p, q, N), markov_u32(p, q, N), markov_u64(p, q, N));
		double m_aa = BY_TYPE(dt, markov_u8(p, p, N), markov_u16(p, p, N), markov_u32(p, p, N), markov_u64(p, p, N));
		double d_ab = log(m_ba / m_bb) / (double)(b->mag - N);
		double d_ba = log(m_ab / m_aa) / (double)(a->mag - N);
		return 1 - exp(0.5 * (d_ab + d_ba));
	}
	default:
		return NAN;
	}
}

/* Feature<T>::feat_is_sim, predict/Feature.cpp:549-663 (in-scope flags) */
int orc_feat_is_sim(uint64_t f) {
	switch (f) {
	case ORC_FEAT_NORMALIZED_VECTORS: case ORC_FEAT_PEARSON_COEFF: case ORC_FEAT_INTERSECTION:
	case ORC_FEAT_KULCZYNSKI2: case ORC_FEAT_SIMRATIO: case ORC_FEAT_SIM_MM:
		return 1;
	case ORC_FEAT_MANHATTAN: case ORC_FEAT_EUCLIDEAN: case ORC_FEAT_JEFFEREY_DIV: case ORC_FEAT_EMD:
	case ORC_FEAT_LENGTHD: case ORC_FEAT_JENSEN_SHANNON: case ORC_FEAT_RRE_K_R:
		return 0;
	default:
		return -1;
	}
}

/* ------------------------------------------------------------------ a5/a7: model */

static int index_of(const orc_model* m, uint64_t f) {
	for (int i = 0; i < m->n_singles; i++) if (m->single_flag[i] == f) return i;
	return -1;
}

/* Feature<T>::add_feature, predict/Feature.cpp:102-128: singles are appended in ascending-bit order of
 * first appearance; the combo remembers the indices in ascending-bit order. */
int orc_model_add_feature(orc_model* m, uint64_t f_flags, int file_code) {
	if (file_code < 0 || file_code > 3 || m->n_combos >= ORC_MAX_COMBOS) return -1;
	int c = m->n_combos;
	m->combo_n[c] = 0;
	for (uint64_t f = 1; f <= f_flags && f != 0; f <<= 1) {
		if ((f_flags & f) != 0) {
			if (index_of(m, f) < 0) {
				int is = orc_feat_is_sim(f);
				if (is < 0 || m->n_singles >= ORC_MAX_SINGLES) return -1;     /* `throw single_flag` */
				int i = m->n_singles++;
				m->single_flag[i] = f;
				m->mins[i] = DBL_MAX;       /* numeric_limits<double>::max() */
				m->maxs[i] = DBL_MIN;       /* numeric_limits<double>::min() (sic) */
				m->is_sim[i] = is;
			}
			m->combo_idx[c][m->combo_n[c]++] = index_of(m, f);
		}
	}
	m->combo_kind[c] = file_code;
	m->combo_flags[c] = f_flags;
	m->n_combos++;
	return 0;
}

int orc_model_set_normal(orc_model* m, uint64_t flag, double mn, double mx) {
	int i = index_of(m, flag);
	if (i < 0) return -1;
	m->mins[i] = mn;
	m->maxs[i] = mx;
	return 0;
}

/* token reader with `in >> x` semantics */
static const char* next_tok(const char* s, char* buf, size_t cap) {
	while (*s && isspace((unsigned char)*s)) s++;
	if (!*s) return NULL;
	size_t n = 0;
	while (*s && !isspace((unsigned char)*s)) { if (n + 1 < cap) buf[n++] = *s; s++; }
	buf[n] = 0;
	return s;
}

/* Predictor<T>::read_from, predict/Predictor.cpp:125-185 */
static const char* read_block(const char* s, int k, orc_model* m) {
	char t[128];
	memset(m, 0, sizeof(*m));
	m->k = k;
	if (!(s = next_tok(s, t, sizeof t))) return NULL;            /* "n_combos:" */
	if (!(s = next_tok(s, t, sizeof t))) return NULL;
	int nc = atoi(t);
	if (nc < 0 || nc > ORC_MAX_COMBOS) return NULL;
	if (!(s = next_tok(s, t, sizeof t))) return NULL;
	m->weights[0] = strtod(t, NULL);
	for (int i = 0; i < nc; i++) {
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		int cmb = atoi(t);
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		uint64_t flags = strtoull(t, NULL, 10);
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		m->weights[i + 1] = strtod(t, NULL);
		if (orc_model_add_feature(m, flags, cmb) != 0) return NULL;
	}
	if (!(s = next_tok(s, t, sizeof t))) return NULL;            /* "n_singles:" */
	if (!(s = next_tok(s, t, sizeof t))) return NULL;
	int ns = atoi(t);
	for (int i = 0; i < ns; i++) {
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		uint64_t f = strtoull(t, NULL, 10);
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		double mn = strtod(t, NULL);
		if (!(s = next_tok(s, t, sizeof t))) return NULL;
		double mx = strtod(t, NULL);
		if (orc_model_set_normal(m, f, mn, mx) != 0) return NULL;
	}
	return s;
}

/* Predictor<T>::Predictor(const std::string filename), predict/Predictor.cpp:47-79 */
int orc_predictor_parse(const char* s, orc_predictor* p) {
	char t[128];
	memset(p, 0, sizeof(*p));
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, t, sizeof t))) return -1;
	p->k = atoi(t);
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, t, sizeof t))) return -1;
	p->mode = atoi(t);
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, t, sizeof t))) return -1;
	p->max_features = atoi(t);
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, t, sizeof t))) return -1;
	p->id = strtod(t, NULL);
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, p->datatype, sizeof p->datatype))) return -1;
	if (!(s = next_tok(s, t, sizeof t)) || !(s = next_tok(s, t, sizeof t))) return -1;
	p->feature_set = strtoull(t, NULL, 10);
	if (p->mode & 1) { if (!(s = read_block(s, p->k, &p->cls))) return -1; }
	if (p->mode & 2) { if (!(s = read_block(s, p->k, &p->reg))) return -1; }
	return 0;
}

int orc_predictor_load(const char* path, orc_predictor* p) {
	FILE* f = fopen(path, "rb");
	if (!f) return -1;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	char* buf = (char*)malloc((size_t)n + 1);
	size_t got = fread(buf, 1, (size_t)n, f);
	buf[got] = 0;
	fclose(f);
	int r = orc_predictor_parse(buf, p);
	free(buf);
	return r;
}

/* `out << std::setprecision(15) << double` == printf("%.15g") */
static int write_block(char* b, size_t cap, const orc_model* m) {
	int n = 0;
	n += snprintf(b + n, cap > (size_t)n ? cap - n : 0, "\nn_combos: %d\n%.15g\n", m->n_combos, m->weights[0]);
	for (int j = 0; j < m->n_combos; j++)
		n += snprintf(b + n, cap > (size_t)n ? cap - n : 0, "%d %llu %.15g\n", m->combo_kind[j],
		              (unsigned long long)m->combo_flags[j], m->weights[j + 1]);
	n += snprintf(b + n, cap > (size_t)n ? cap - n : 0, "\nn_singles: %d\n", m->n_singles);
	for (int j = 0; j < m->n_singles; j++)
		n += snprintf(b + n, cap > (size_t)n ? cap - n : 0, "%llu %.15g %.15g\n", (unsigned long long)m->single_flag[j], m->mins[j], m->maxs[j]);
	return n;
}

/* Predictor<T>::save + write_to, predict/Predictor.cpp:28-44,82-121 */
int orc_predictor_format(const orc_predictor* p, char* buf, size_t cap) {
	int n = 0;
	n += snprintf(buf + n, cap > (size_t)n ? cap - n : 0, "k: %d\nmode: %u\nmax_features: %d\nID: %g\nDatatype: %s\nfeature_set: %llu\n",
	              p->k, (unsigned)p->mode, p->max_features, p->id, p->datatype, (unsigned long long)p->feature_set);
	if (p->mode & 1) n += write_block(buf + n, cap > (size_t)n ? cap - n : 0, &p->cls);
	if (p->mode & 2) n += write_block(buf + n, cap > (size_t)n ? cap - n : 0, &p->reg);
	return n;
}

/* Feature<T>::compute = compute_all_raw + normalize_cache, predict/Feature.h:197-201, Feature.cpp:137-171.
 * One full pass over both histograms PER single feature, like the reference. */
int orc_compute(const orc_model* m, const orc_hist* a, const orc_hist* b, double* singles) {
	for (int i = 0; i < m->n_singles; i++) {
		int err = 0;
		singles[i] = orc_raw_feature(m->single_flag[i], a, b, &err);
		if (err) return -err;
	}
	for (int i = 0; i < m->n_singles; i++) {
		double val = (singles[i] - m->mins[i]) / (m->maxs[i] - m->mins[i]);
		if (isnan(val)) return -1;                      /* `throw std::exception()`, Feature.cpp:143-146 */
		singles[i] = m->is_sim[i] ? val : 1 - val;
	}
	return 0;
}

/* Feature<T>::operator()(col, cache), predict/Feature.h:205-239 */
double orc_combo(const orc_model* m, int col, const double* c) {
	const int* idx = m->combo_idx[col];
	const int n = m->combo_n[col];
	switch (m->combo_kind[col]) {
	case 0: { double prod = 1; for (int i = 0; i < n; i++) prod *= c[idx[i]]; return prod; }                 /* xy */
	case 3: { double prod = 1; for (int i = 0; i < n; i++) prod *= c[idx[i]] * c[idx[i]]; return prod; }     /* x2y2 */
	case 1: return n == 2 ? c[idx[0]] * c[idx[1]] * c[idx[1]] : NAN;                                         /* xy2 */
	case 2: return n == 2 ? c[idx[0]] * c[idx[0]] * c[idx[1]] : NAN;                                         /* x2y */
	default: return NAN;
	}
}

double orc_weighted_sum(const orc_model* m, const double* singles) {
	double sum = m->weights[0];
	for (int col = 1; col <= m->n_combos; col++) sum += m->weights[col] * orc_combo(m, col - 1, singles);
	return sum;
}

double orc_logistic(double x) { return 1.0 / (1 + exp(-x)); }

/* Trainer<T>::classify, cluster/Trainer.cpp:112-120 */
double orc_classify(const orc_predictor* p, const orc_hist* a, const orc_hist* b) {
	double singles[ORC_MAX_SINGLES];
	if (orc_compute(&p->cls, a, b, singles) != 0) return NAN;
	return orc_logistic(orc_weighted_sum(&p->cls, singles)) + p->bias;
}

int orc_p_close(const orc_predictor* p, const orc_hist* a, const orc_hist* b) {
	return round(orc_classify(p, a, b)) > 0;
}

double orc_p_predict(const orc_predictor* p, const orc_hist* a, const orc_hist* b) {
	double singles[ORC_MAX_SINGLES];
	if (orc_compute(&p->reg, a, b, singles) != 0) return NAN;
	double sum = orc_weighted_sum(&p->reg, singles);
	if (sum < 0) sum = 0; else if (sum > 1) sum = 1;
	return sum;
}

/* ------------------------------------------------------------------ a8/a9: Trainer operators */

static double get_id(double cutoff) { return cutoff > 1 ? cutoff / 100.0 : cutoff; }   /* Trainer.h:35 */

typedef struct { int64_t pos; double sim; } best_t;

static inline int close_one(const orc_predictor* p, const orc_hist* pt, const orc_hist* q, double* dist) {
	double singles[ORC_MAX_SINGLES];
	/* auto cache = feat->compute(*pt, *p); double dist = (*feat)(0, cache);  Trainer.cpp:49-50 */
	if (orc_compute(&p->cls, pt, q, singles) != 0) return -1;
	*dist = orc_combo(&p->cls, 0, singles);
	/* double sum = classify(pt, p)  -- computes everything a second time, Trainer.cpp:51 */
	double sum = orc_classify(p, pt, q);
	return round(sum) > 0 ? 1 : 0;
}

int orc_get_close(const orc_predictor* p, double cutoff, const orc_hist* query, const orc_hist* const* cands, size_t m,
                  uint8_t* flags, int64_t* best_pos, double* best_sim, int* is_min_out) {
	best_t best = {-1, -1};
	int is_min = 1;
	uint64_t min_len = (uint64_t)(query->length * cutoff);     /* Trainer.cpp:39-40 (raw cutoff, not get_id()) */
	uint64_t max_len = (uint64_t)(query->length / cutoff);
	for (size_t i = 0; i < m; i++) {
		flags[i] = 0;
		uint64_t len = cands[i]->length;
		if (len < min_len || len > max_len) continue;
		double dist;
		int res = close_one(p, cands[i], query, &dist);
		if (res < 0) return -1;
		if (dist > best.sim) { best.pos = (int64_t)i; best.sim = dist; }
		is_min = is_min && (res != 1);
		if (res == 1) flags[i] = 1;
	}
	*best_pos = best.pos;
	*best_sim = best.sim;
	*is_min_out = is_min;
	return 0;
}

int orc_get_close_omp(const orc_predictor* p, double cutoff, const orc_hist* query, const orc_hist* const* cands, size_t m,
                      uint8_t* flags, int64_t* best_pos, double* best_sim, int* is_min_out) {
	uint64_t min_len = (uint64_t)(query->length * cutoff);
	uint64_t max_len = (uint64_t)(query->length / cutoff);
	double* sims = (double*)malloc(sizeof(double) * (m ? m : 1));
	int bad = 0;
#pragma omp parallel for schedule(static) reduction(|:bad)
	for (size_t i = 0; i < m; i++) {
		flags[i] = 0;
		sims[i] = -INFINITY;
		uint64_t len = cands[i]->length;
		if (len < min_len || len > max_len) continue;
		double dist;
		int res = close_one(p, cands[i], query, &dist);
		if (res < 0) { bad = 1; continue; }
		sims[i] = dist;
		if (res == 1) flags[i] = 1;
	}
	best_t best = {-1, -1};
	int is_min = 1;
	for (size_t i = 0; i < m; i++) {      /* serial order of the pmax / && reductions */
		if (sims[i] == -INFINITY) continue;
		if (sims[i] > best.sim) { best.pos = (int64_t)i; best.sim = sims[i]; }
		if (flags[i]) is_min = 0;
	}
	free(sims);
	*best_pos = best.pos;
	*best_sim = best.sim;
	*is_min_out = is_min;
	return bad ? -1 : 0;
}

/* Trainer<T>::filter, cluster/Trainer.cpp:123-141: classify(centre, point) */
int orc_filter(const orc_predictor* p, double cutoff, const orc_hist* centre, const orc_hist* const* pts, size_t m, uint8_t* keep) {
	uint64_t cen_length = centre->length;
	uint64_t min_length = (uint64_t)(cen_length * get_id(cutoff));
	uint64_t max_length = (uint64_t)(cen_length / get_id(cutoff));
	int kept = 0;
	for (size_t i = 0; i < m; i++) {
		uint64_t l = pts[i]->length;
		int length_pass = l >= min_length && l <= max_length;
		int remove = 1;
		if (length_pass) {
			double sum = orc_classify(p, centre, pts[i]);
			double res = round(sum);
			remove = (res == 0);
		}
		keep[i] = !remove;
		kept += !remove;
	}
	return kept;
}

/* Trainer<T>::merge, cluster/Trainer.cpp:74-109: compute(*cen_i, *current) */
long orc_merge(const orc_predictor* p, double cutoff, const orc_hist* const* centres, size_t n, long current, long begin, long last) {
	(void)n;
	long best_i = 0;
	double best_d = DBL_MIN;                                   /* numeric_limits<double>::min() */
	const orc_hist* cur = centres[current];
	uint64_t cen_length = cur->length;
	uint64_t min_length = (uint64_t)(cen_length * get_id(cutoff));
	uint64_t max_length = (uint64_t)(cen_length / get_id(cutoff));
	for (long i = begin; i <= last; i++) {
		const orc_hist* cen = centres[i];
		uint64_t cl = cen->length;
		if (cl >= min_length && cl <= max_length) {
			double singles[ORC_MAX_SINGLES];
			if (orc_compute(&p->cls, cen, cur, singles) != 0) return -2;
			double dist = orc_combo(&p->cls, 0, singles);
			double sum = orc_weighted_sum(&p->cls, singles);
			double res = round(orc_logistic(sum) + p->bias);
			if (res == 1) {
				if (!(best_d > dist)) { best_i = i; best_d = dist; }
			}
		}
	}
	return best_i;
}

/* ------------------------------------------------------------------ a4/a10 */

double orc_distance_d(const orc_hist* a, const double* mean) {
	return BY_TYPE(a->dtype, distance_d_u8((const uint8_t*)a->bins, mean, a->nbins), distance_d_u16((const uint16_t*)a->bins, mean, a->nbins),
	               distance_d_u32((const uint32_t*)a->bins, mean, a->nbins), distance_d_u64((const uint64_t*)a->bins, mean, a->nbins));
}

uint64_t orc_distance(const orc_hist* a, const orc_hist* b) {
	return BY_TYPE(a->dtype, distance_u8(a->bins, b->bins, a->nbins, a->mag, b->mag), distance_u16(a->bins, b->bins, a->nbins, a->mag, b->mag),
	               distance_u32(a->bins, b->bins, a->nbins, a->mag, b->mag), distance_u64(a->bins, b->bins, a->nbins, a->mag, b->mag));
}

int orc_mean_nearest(const orc_hist* const* pts, size_t m, double* mean_out, double* dists, int64_t* nearest) {
	if (m == 0) return -1;
	const size_t N = pts[0]->nbins;
	double* top = (double*)calloc(N, sizeof(double));
	for (size_t i = 0; i < m; i++) {
		switch (pts[i]->dtype) {
		case 8: add_to_mean_u8((const uint8_t*)pts[i]->bins, top, N); break;
		case 16: add_to_mean_u16((const uint16_t*)pts[i]->bins, top, N); break;
		case 32: add_to_mean_u32((const uint32_t*)pts[i]->bins, top, N); break;
		default: add_to_mean_u64((const uint64_t*)pts[i]->bins, top, N); break;
		}
	}
	double bottom = (double)m;
	for (size_t i = 0; i < N; i++) top[i] /= bottom;           /* *top /= bottom, ClusterFactory.cpp:357 */
	int64_t best = -1;
	double bd = 0;
	for (size_t i = 0; i < m; i++) {
		double d = orc_distance_d(pts[i], top);
		if (dists) dists[i] = d;
		if (best < 0 || d < bd) { bd = d; best = (int64_t)i; }     /* first minimum wins */
	}
	if (mean_out) memcpy(mean_out, top, N * sizeof(double));
	*nearest = best;
	free(top);
	return 0;
}

void orc_set_threads(int n) {
#ifdef _OPENMP
	omp_set_num_threads(n);
#else
	(void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
