// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" glue over the REAL MeShClust2 reference, compiled from its own sources where
// they lie under /root/reference (see oracle/Makefile target `ref`). It contains no algorithm of
// its own: every function below forwards to the reference symbol named in its comment. It is used
//   (1) to pin oracle/msc_oracle.c (the travelling C restatement) against the real code, and
//   (2) by tests/golden/gen_golden.py to produce the committed golden fixtures.
// It only exists in this container (the GPU box has no /root/reference); the built
// oracle/_ref/libmsc_ref.so travels with the snapshot and is optional there.
//
// Private members of Trainer/Predictor/Feature are reached by the usual test-only trick of
// re-defining `private` AFTER all standard headers have been included.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <numeric>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>
#include <omp.h>

#define private public
#define protected public
#include "clutil/Loader.h"
#include "clutil/DivergencePoint.h"
#include "clutil/Datatype.h"
#include "predict/Feature.h"
#include "predict/Predictor.h"
#include "predict/GLM.h"
#include "predict/BestFirstSelector.h"
#include "cluster/Trainer.h"
#include "cluster/bvec.h"
#include "cluster/bvec_iterator.h"
#include "cluster/Center.h"
#undef private
#undef protected

namespace {

template <class T> struct RefPoint {          // owns one reference Point<T>
	Point<T>* p;
};

// nonltr/ChromosomeOneDigitDna + clutil/Loader.cpp:138-179 (chromosome overload) or
// clutil/Loader.cpp:112-134 (string overload, strips non-ACGT first) when strip != 0.
template <class T> void* point_create(const char* header, const char* seq, int k, int strip) {
	static uintmax_t next_id = 0;
	uintmax_t id = next_id++;
	std::string h(header), s(seq);
	Point<T>* p = NULL;
	if (strip) {
		p = Loader<T>::get_point(h, s, id, k, false);
	} else {
		ChromosomeOneDigitDna* chrom = new ChromosomeOneDigitDna();
		chrom->setHeader(h);
		chrom->appendToSequence(s);
		chrom->finalize();
		p = Loader<T>::get_point(chrom, id, k, false);
		delete chrom;
	}
	return (void*)p;
}

template <class T> DivergencePoint<T>* dp(void* h) {
	return dynamic_cast<DivergencePoint<T>*>((Point<T>*)h);
}

template <class T> int point_bins(void* h, void* out) {
	auto d = dp<T>(h);
	std::memcpy(out, d->points.data(), d->points.size() * sizeof(T));
	return (int)d->points.size();
}

template <class T> void point_meta(void* h, uint64_t* mag, uint64_t* len, double* stddev, uint64_t* one_mers) {
	auto d = dp<T>(h);
	*mag = d->getPseudoMagnitude();
	*len = d->get_length();
	*stddev = d->get_stddev();
	auto om = d->get_1mers();
	for (int i = 0; i < 4; i++) one_mers[i] = i < (int)om.size() ? om[i] : 0;   // clone() does not carry the 1-mers
}

// One raw statistic, straight from predict/Feature.cpp (static members; jensen_shannon and the
// n2r* family need an instance).
template <class T> double raw_feature(uint64_t flag, void* a, void* b, int k) {
	Point<T>& p = *(Point<T>*)a;
	Point<T>& q = *(Point<T>*)b;
	switch (flag) {
	case FEAT_MANHATTAN: return Feature<T>::manhattan(p, q);
	case FEAT_EUCLIDEAN: return Feature<T>::euclidean(p, q);
	case FEAT_NORMALIZED_VECTORS: return Feature<T>::normalized_vectors(p, q);
	case FEAT_JEFFEREY_DIV: return Feature<T>::jefferey_divergence(p, q);
	case FEAT_PEARSON_COEFF: return Feature<T>::pearson(p, q);
	case FEAT_INTERSECTION: return Feature<T>::intersection(p, q);
	case FEAT_EMD: return Feature<T>::emd(p, q);
	case FEAT_LENGTHD: return Feature<T>::length_difference(p, q);
	case FEAT_KULCZYNSKI2: return Feature<T>::kulczynski2(p, q);
	case FEAT_SIMRATIO: return Feature<T>::simratio(p, q);
	case FEAT_JENSEN_SHANNON: {
		Feature<T> f(k);
		return f.jensen_shannon(p, q);
	}
	case FEAT_RRE_K_R: return Feature<T>::rre_k_r(p, q);        // predict/Feature.cpp:1029-1062
	case FEAT_SIM_MM: return Feature<T>::sim_mm(p, q);          // :1451-1455
	case FEAT_MARKOV: return Feature<T>::markov(p, q);          // :1367-1393
	default: return std::nan("");
	}
}

template <class T> struct RefModel {
	Predictor<T>* pred;      // leaked on purpose (~Predictor on a file-loaded object is UB)
	Feature<T>* feat;        // classification feature set (copy, do_save=false) -- Predictor.h:57
	matrix::Matrix weights;
	Feature<T>* feat_r;      // regression block if present
	matrix::Matrix weights_r;
	bool has_r;
};

// predict/Predictor.cpp:47-79 (file ctor) -> read_from :125-185
template <class T> void* model_load(const char* path) {
	RefModel<T>* m = new RefModel<T>();
	m->pred = new Predictor<T>(std::string(path));
	// Feature(int k) leaves do_save uninitialised (predict/Feature.h:111-153); with it set the raw statistics are
	// memoised by point id. The clustering path always works on a do_save=false copy (Predictor.h:57), so do the same.
	if (m->pred->feat_c) m->pred->feat_c->set_save(false);
	if (m->pred->mode & PRED_MODE_REGR) m->pred->feat_r->set_save(false);
	auto pr = m->pred->get_class();
	m->feat = pr.first;
	m->feat->set_save(false);
	m->weights = pr.second.get_weights();
	m->has_r = (m->pred->mode & PRED_MODE_REGR) != 0;
	if (m->has_r) {
		m->feat_r = new Feature<T>(*m->pred->feat_r);
		m->feat_r->set_save(false);
		m->weights_r = m->pred->r_glm.get_weights();
	}
	return m;
}

// Feature::compute (predict/Feature.h:197-201) then operator() per combo (:205-239) and the
// weighted sum exactly as Trainer::classify does it (cluster/Trainer.cpp:112-120).
template <class T>
int model_score(void* mh, void* a, void* b, double* singles, double* combos, double* sum, double* csum) {
	RefModel<T>* m = (RefModel<T>*)mh;
	auto cache = m->feat->compute(*(Point<T>*)a, *(Point<T>*)b);
	for (size_t i = 0; i < cache.size(); i++) singles[i] = cache[i];
	double s = m->weights.get(0, 0);
	for (int col = 1; col < m->weights.getNumRow(); col++) {
		double d = (*m->feat)(col - 1, cache);
		combos[col - 1] = d;
		s += m->weights.get(col, 0) * d;
	}
	*sum = s;
	*csum = Predictor<T>::classify_sum(s);
	return (int)cache.size();
}

// predict/Predictor.cpp:284-300 p_predict (regression block)
template <class T> double model_predict(void* mh, void* a, void* b) {
	RefModel<T>* m = (RefModel<T>*)mh;
	return m->pred->p_predict((Point<T>*)a, (Point<T>*)b);
}
template <class T> int model_close(void* mh, void* a, void* b) {
	RefModel<T>* m = (RefModel<T>*)mh;
	return m->pred->p_close((Point<T>*)a, (Point<T>*)b) ? 1 : 0;
}

template <class T> Trainer<T>* make_trainer(RefModel<T>* m, double cutoff) {
	std::vector<Point<T>*> none;
	Trainer<T>* t = new Trainer<T>(none, 0, 0, cutoff, 0, m->feat->get_k());
	delete t->feat;
	t->feat = new Feature<T>(*m->feat);
	t->feat->set_save(false);
	t->weights = m->weights;
	return t;
}

// cluster/Trainer.cpp:23-71. Candidates are laid out in ONE bvec row; [0, m) is the window.
template <class T>
int get_close(void* mh, double cutoff, void* query, void** cands, int m, uint8_t* flags,
              int64_t* best_pos, double* best_sim, int* is_min) {
	RefModel<T>* mod = (RefModel<T>*)mh;
	Trainer<T>* t = make_trainer<T>(mod, cutoff);
	typename bvec_iterator<T>::vtype col(1);
	for (int i = 0; i < m; i++) col[0].push_back(std::make_pair((Point<T>*)cands[i], false));
	bvec_iterator<T> b(0, 0, &col), e(0, (size_t)m, &col);
	bool ismin = false;
	auto res = t->get_close((Point<T>*)query, b, e, ismin);
	for (int i = 0; i < m; i++) flags[i] = col[0][i].second ? 1 : 0;
	*is_min = ismin ? 1 : 0;
	*best_sim = std::get<1>(res);
	*best_pos = std::get<0>(res) == NULL ? -1 : (int64_t)std::get<3>(res);
	delete t;
	return 0;
}

// cluster/Trainer.cpp:123-141: keep[i] = 1 iff the point survives filter()
template <class T>
int filter(void* mh, double cutoff, void* centre, void** pts, int m, uint8_t* keep) {
	RefModel<T>* mod = (RefModel<T>*)mh;
	Trainer<T>* t = make_trainer<T>(mod, cutoff);
	std::vector<std::pair<Point<T>*, bool> > vec;
	for (int i = 0; i < m; i++) vec.push_back(std::make_pair((Point<T>*)pts[i], false));
	t->filter((Point<T>*)centre, vec);
	std::set<Point<T>*> kept;
	for (auto& pr : vec) kept.insert(pr.first);
	for (int i = 0; i < m; i++) keep[i] = kept.count((Point<T>*)pts[i]) ? 1 : 0;
	delete t;
	return (int)vec.size();
}

// cluster/Trainer.cpp:74-109
template <class T>
long merge(void* mh, double cutoff, void** centres, int n, long current, long begin, long last) {
	RefModel<T>* mod = (RefModel<T>*)mh;
	Trainer<T>* t = make_trainer<T>(mod, cutoff);
	std::vector<Center<T> > cs;
	std::vector<Point<T>*> none;
	for (int i = 0; i < n; i++) cs.push_back(Center<T>((Point<T>*)centres[i], none));
	long r = t->merge(cs, current, begin, last);
	delete t;
	return r;
}

// Column mean (cluster/ClusterFactory.cpp:338-356) + distance_d to it for every member
// (clutil/DivergencePoint.cpp:55-66) + first arg-min (cluster/Trainer.cpp:144-157).
template <class T>
int mean_nearest(void** pts, int m, double* mean_out, double* dists, int64_t* nearest) {
	Point<T>* first = (Point<T>*)pts[0];
	Point<double>* top = first->create_double();
	top->zero();
	Point<double>* temp = top->clone();
	for (int i = 0; i < m; i++) {
		((Point<T>*)pts[i])->set_arg_to_this_d(*temp);
		*top += *temp;
	}
	*top /= (double)m;
	auto dtop = dynamic_cast<DivergencePoint<double>*>(top);
	if (mean_out) std::memcpy(mean_out, dtop->points.data(), dtop->points.size() * sizeof(double));
	int64_t best = -1;
	double bd = 0;
	for (int i = 0; i < m; i++) {
		double d = ((Point<T>*)pts[i])->distance_d(*top);
		dists[i] = d;
		if (best < 0 || d < bd) { bd = d; best = i; }
	}
	*nearest = best;
	delete top;
	delete temp;
	return 0;
}

template <class T> uint64_t distance(void* a, void* b) {
	return ((Point<T>*)a)->distance(*(Point<T>*)b);
}

// DivergencePoint::set (clutil/DivergencePoint.cpp:182-190): bins/len/header/id copied, mag NOT.
template <class T> void* clone_point(void* a) { return (void*)((Point<T>*)a)->clone(); }
template <class T> void set_point(void* dst, void* src) { ((Point<T>*)dst)->set(*(Point<T>*)src); }
template <class T> void free_point(void* a) { delete (Point<T>*)a; }

// Predictor<T>::train() + train_class (predict/Predictor.cpp:876-975) on caller-supplied labelled pairs instead of the mutants
// the reference generates itself: the single features of `feat_flags` are added and normalised over training, then testing
// (:889-894), the candidate list is what Predictor<T>::add_feats enumerates (:201-220, forwarded to through a bare object),
// BestFirstSelector<T>::train_class picks the combos and fits the GLM (predict/BestFirstSelector.cpp:187-250), and
// Predictor<T>::write_to prints the block (:82-121). Returns the text length, or -1.
template <class T>
long train_class(void** first, void** second, const double* val, int n_train, int n_test, int k, uint64_t feat_flags, int min_feat, int max_feat,
                 double id, char* out, long cap, double* acc_out) {
	std::vector<pra<T> > training, testing;
	for (int i = 0; i < n_train + n_test; i++) {
		pra<T> pr((Point<T>*)first[i], (Point<T>*)second[i], val[i]);
		(i < n_train ? training : testing).push_back(pr);
	}
	std::vector<std::pair<uint64_t, Combo> > possible;
	Predictor<T>* bare = static_cast<Predictor<T>*>(::operator new(sizeof(Predictor<T>)));      // add_feats touches no member
	bare->add_feats(possible, feat_flags);
	::operator delete(bare);
	Feature<T> feat(k);
	feat.set_save(true);
	for (uint64_t i = 1; i <= feat_flags; i *= 2) if (i & feat_flags) feat.add_feature(i, Combo::xy);
	feat.normalize(training);
	feat.normalize(testing);
	feat.finalize();
	BestFirstSelector<T> sel(possible, min_feat, max_feat);
	auto pr = sel.train_class(&feat, training, testing, id);
	if (acc_out) {
		auto tr = FeatureSelector<T>::class_test(training, *pr.first, pr.second, id);
		auto te = FeatureSelector<T>::class_test(testing, *pr.first, pr.second, id);
		acc_out[0] = std::get<0>(tr);
		acc_out[1] = std::get<0>(te);
	}
	const std::string path = "/tmp/msc_ref_train_" + std::to_string((long)omp_get_wtime()) + "_" + std::to_string((long)(uintptr_t)out) + ".txt";
	{
		std::ofstream ofs(path);
		Predictor<T>* writer = static_cast<Predictor<T>*>(::operator new(sizeof(Predictor<T>)));  // write_to touches no member either
		writer->write_to(ofs, pr.first, pr.second);
		::operator delete(writer);
	}
	std::ifstream ifs(path);
	std::stringstream ss;
	ss << ifs.rdbuf();
	std::remove(path.c_str());
	const std::string text = ss.str();
	if ((long)text.size() + 1 > cap) return -1;
	memcpy(out, text.c_str(), text.size() + 1);
	return (long)text.size();
}

// Predictor<T>::train_regr (predict/Predictor.cpp:977-985) on caller-supplied labelled pairs. It hands the work to
// GreedySelector<T>::train_regression (predict/GreedySelector.cpp:11-76), which cannot be CALLED: the function falls off its end
// without a return statement (its caller then destroys a pair<Feature*, GLM> that was never written -- the std::bad_cast /
// crash of `fastcar --dump`). So its body is followed here step by step on the reference's own objects and primitives --
// Feature<T>::remove_feature / add_feature / normalize / finalize, FeatureSelector<T>::regression_train / regression_test
// (predict/FeatureSelector.cpp:41-55,77-89), the candidate list of Predictor<T>::add_feats -- and what its last lines compute
// (feat_r, r_glm) is printed by Predictor<T>::write_to. Returns the text length, or -1; err_out = {training, testing} mean errors.
template <class T>
long train_regr(void** first, void** second, const double* val, int n_train, int n_test, int k, uint64_t feat_flags, int max_feat, char* out, long cap,
                double* err_out) {
	std::vector<pra<T> > training, testing;
	for (int i = 0; i < n_train + n_test; i++) {
		pra<T> pr((Point<T>*)first[i], (Point<T>*)second[i], val[i]);
		(i < n_train ? training : testing).push_back(pr);
	}
	std::vector<std::pair<uint64_t, Combo> > possible_feats;
	Predictor<T>* bare = static_cast<Predictor<T>*>(::operator new(sizeof(Predictor<T>)));      // add_feats touches no member
	bare->add_feats(possible_feats, feat_flags);
	::operator delete(bare);
	Feature<T> feat_obj(k);
	Feature<T>* feat = &feat_obj;
	feat->set_save(true);
	for (uint64_t i = 1; i <= feat_flags; i *= 2) if (i & feat_flags) feat->add_feature(i, Combo::xy);      // Predictor<T>::train, :881-894
	feat->normalize(training);
	feat->normalize(testing);
	feat->finalize();
	// ---- GreedySelector<T>::train_regression, :14-52
	const int max_num_feat = max_feat;
	auto c_size = feat->get_combos().size();
	for (size_t i = 0; i < c_size; i++) feat->remove_feature();
	std::vector<uintmax_t> used_list;
	double abs_best_regr = 1000000;
	for (auto num_feat = 1; num_feat <= max_num_feat; num_feat++) {
		double best_regr_err = abs_best_regr;
		uintmax_t best_idx = -1;
		auto best_regr_feat = possible_feats.front();
		for (uint64_t i = 0; i < possible_feats.size(); i++) {
			if (std::find(used_list.begin(), used_list.end(), i) != used_list.end()) continue;
			auto rfeat = possible_feats[i];
			feat->add_feature(rfeat.first, rfeat.second);
			feat->normalize(training);
			feat->finalize();
			auto pr = FeatureSelector<T>::regression_train(training, *feat);
			double regr_mse = FeatureSelector<T>::regression_test(testing, *feat, pr.second);
			feat->remove_feature();
			if (regr_mse < best_regr_err) { best_regr_err = regr_mse; best_regr_feat = rfeat; best_idx = i; }
		}
		if (best_regr_err < abs_best_regr) {
			feat->add_feature(best_regr_feat.first, best_regr_feat.second);
			feat->normalize(training);
			feat->finalize();
			abs_best_regr = best_regr_err;
			used_list.push_back(best_idx);
		}
	}
	Feature<T>* feat_r = new Feature<T>(*feat);      // :55-58
	feat_r->set_save(false);
	auto pr = FeatureSelector<T>::regression_train(training, *feat_r);
	matrix::GLM r_glm = pr.second;
	if (err_out) { err_out[0] = pr.first; err_out[1] = FeatureSelector<T>::regression_test(testing, *feat_r, r_glm); }
	const std::string path = "/tmp/msc_ref_regr_" + std::to_string((long)omp_get_wtime()) + "_" + std::to_string((long)(uintptr_t)out) + ".txt";
	{
		std::ofstream ofs(path);
		Predictor<T>* writer = static_cast<Predictor<T>*>(::operator new(sizeof(Predictor<T>)));
		writer->write_to(ofs, feat_r, r_glm);
		::operator delete(writer);
	}
	std::ifstream ifs(path);
	std::stringstream ss;
	ss << ifs.rdbuf();
	std::remove(path.c_str());
	const std::string text = ss.str();
	if ((long)text.size() + 1 > cap) return -1;
	memcpy(out, text.c_str(), text.size() + 1);
	return (long)text.size();
}

}  // namespace

#define DISPATCH(dtype, expr8, expr16, expr32, expr64) \
	switch (dtype) { case 8: expr8; break; case 16: expr16; break; case 32: expr32; break; case 64: expr64; break; default: break; }

extern "C" {

void* ref_point_create(int dtype, const char* header, const char* seq, int k, int strip) {
	try {
		DISPATCH(dtype, return point_create<uint8_t>(header, seq, k, strip), return point_create<uint16_t>(header, seq, k, strip),
		         return point_create<uint32_t>(header, seq, k, strip), return point_create<uint64_t>(header, seq, k, strip));
	} catch (...) { return NULL; }
	return NULL;
}
int ref_point_bins(int dtype, void* h, void* out) {
	DISPATCH(dtype, return point_bins<uint8_t>(h, out), return point_bins<uint16_t>(h, out),
	         return point_bins<uint32_t>(h, out), return point_bins<uint64_t>(h, out));
	return -1;
}
void ref_point_meta(int dtype, void* h, uint64_t* mag, uint64_t* len, double* stddev, uint64_t* one_mers) {
	DISPATCH(dtype, point_meta<uint8_t>(h, mag, len, stddev, one_mers), point_meta<uint16_t>(h, mag, len, stddev, one_mers),
	         point_meta<uint32_t>(h, mag, len, stddev, one_mers), point_meta<uint64_t>(h, mag, len, stddev, one_mers));
}
void* ref_point_clone(int dtype, void* a) {
	DISPATCH(dtype, return clone_point<uint8_t>(a), return clone_point<uint16_t>(a), return clone_point<uint32_t>(a), return clone_point<uint64_t>(a));
	return NULL;
}
void ref_point_set(int dtype, void* dst, void* src) {
	DISPATCH(dtype, set_point<uint8_t>(dst, src), set_point<uint16_t>(dst, src), set_point<uint32_t>(dst, src), set_point<uint64_t>(dst, src));
}
void ref_point_free(int dtype, void* a) {
	DISPATCH(dtype, free_point<uint8_t>(a), free_point<uint16_t>(a), free_point<uint32_t>(a), free_point<uint64_t>(a));
}

// Encoding only: nonltr/Chromosome.cpp:130-154 + nonltr/ChromosomeOneDigit.cpp:79-133.
// codes_out gets the in-place encoded string (same length as seq), segs_out pairs [s,e].
int ref_encode(const char* header, const char* seq, char* codes_out, int* segs_out, int max_segs, uint64_t* eff_len) {
	try {
		std::string h(header), s(seq);
		ChromosomeOneDigitDna chrom;
		chrom.setHeader(h);
		chrom.appendToSequence(s);
		chrom.finalize();
		const std::string* b = chrom.getBase();
		std::memcpy(codes_out, b->data(), b->size());
		auto seg = chrom.getSegment();
		int n = 0;
		for (auto v : *seg) {
			if (n < max_segs) { segs_out[2 * n] = v->at(0); segs_out[2 * n + 1] = v->at(1); }
			n++;
		}
		*eff_len = chrom.getEffectiveSize();
		return n;
	} catch (...) { return -1; }
}

double ref_raw_feature(int dtype, uint64_t flag, void* a, void* b, int k) {
	try {
		DISPATCH(dtype, return raw_feature<uint8_t>(flag, a, b, k), return raw_feature<uint16_t>(flag, a, b, k),
		         return raw_feature<uint32_t>(flag, a, b, k), return raw_feature<uint64_t>(flag, a, b, k));
	} catch (...) { return std::nan(""); }
	return std::nan("");
}
uint64_t ref_distance(int dtype, void* a, void* b) {
	DISPATCH(dtype, return distance<uint8_t>(a, b), return distance<uint16_t>(a, b), return distance<uint32_t>(a, b), return distance<uint64_t>(a, b));
	return 0;
}

void* ref_model_load(int dtype, const char* path) {
	try {
		DISPATCH(dtype, return model_load<uint8_t>(path), return model_load<uint16_t>(path), return model_load<uint32_t>(path), return model_load<uint64_t>(path));
	} catch (...) { return NULL; }
	return NULL;
}
int ref_model_score(int dtype, void* m, void* a, void* b, double* singles, double* combos, double* sum, double* csum) {
	try {
		DISPATCH(dtype, return model_score<uint8_t>(m, a, b, singles, combos, sum, csum), return model_score<uint16_t>(m, a, b, singles, combos, sum, csum),
		         return model_score<uint32_t>(m, a, b, singles, combos, sum, csum), return model_score<uint64_t>(m, a, b, singles, combos, sum, csum));
	} catch (...) { return -1; }
	return -1;
}
double ref_model_predict(int dtype, void* m, void* a, void* b) {
	DISPATCH(dtype, return model_predict<uint8_t>(m, a, b), return model_predict<uint16_t>(m, a, b), return model_predict<uint32_t>(m, a, b), return model_predict<uint64_t>(m, a, b));
	return 0;
}
int ref_model_close(int dtype, void* m, void* a, void* b) {
	DISPATCH(dtype, return model_close<uint8_t>(m, a, b), return model_close<uint16_t>(m, a, b), return model_close<uint32_t>(m, a, b), return model_close<uint64_t>(m, a, b));
	return 0;
}
void ref_set_bias(double b) { Predictor<uint32_t>::set_bias(b); }   // one global, predict/Predictor.cpp:307

int ref_get_close(int dtype, void* m, double cutoff, void* query, void** cands, int n, uint8_t* flags,
                  int64_t* best_pos, double* best_sim, int* is_min) {
	try {
		DISPATCH(dtype, return get_close<uint8_t>(m, cutoff, query, cands, n, flags, best_pos, best_sim, is_min),
		         return get_close<uint16_t>(m, cutoff, query, cands, n, flags, best_pos, best_sim, is_min),
		         return get_close<uint32_t>(m, cutoff, query, cands, n, flags, best_pos, best_sim, is_min),
		         return get_close<uint64_t>(m, cutoff, query, cands, n, flags, best_pos, best_sim, is_min));
	} catch (...) { return -1; }
	return -1;
}
int ref_filter(int dtype, void* m, double cutoff, void* centre, void** pts, int n, uint8_t* keep) {
	try {
		DISPATCH(dtype, return filter<uint8_t>(m, cutoff, centre, pts, n, keep), return filter<uint16_t>(m, cutoff, centre, pts, n, keep),
		         return filter<uint32_t>(m, cutoff, centre, pts, n, keep), return filter<uint64_t>(m, cutoff, centre, pts, n, keep));
	} catch (...) { return -1; }
	return -1;
}
long ref_merge(int dtype, void* m, double cutoff, void** centres, int n, long current, long begin, long last) {
	try {
		DISPATCH(dtype, return merge<uint8_t>(m, cutoff, centres, n, current, begin, last), return merge<uint16_t>(m, cutoff, centres, n, current, begin, last),
		         return merge<uint32_t>(m, cutoff, centres, n, current, begin, last), return merge<uint64_t>(m, cutoff, centres, n, current, begin, last));
	} catch (...) { return -2; }
	return -2;
}
int ref_mean_nearest(int dtype, void** pts, int n, double* mean_out, double* dists, int64_t* nearest) {
	try {
		DISPATCH(dtype, return mean_nearest<uint8_t>(pts, n, mean_out, dists, nearest), return mean_nearest<uint16_t>(pts, n, mean_out, dists, nearest),
		         return mean_nearest<uint32_t>(pts, n, mean_out, dists, nearest), return mean_nearest<uint64_t>(pts, n, mean_out, dists, nearest));
	} catch (...) { return -1; }
	return -1;
}

void ref_set_threads(int n) { omp_set_num_threads(n); }

// ---- cluster/bvec.{h,cpp} + cluster/bvec_iterator.h on their own: the same entry points as msc_bins_* of libmsc_driver.so
// (include/meshclust2_driver.h), each forwarding to the reference's bvec<uint8_t>. Points are 1-bin DivergencePoints that only
// carry a length and an id.
struct RefBins {
	bvec<uint8_t>* bv;
	std::vector<Point<uint8_t>*> pts;
};
void* ref_bins_create(const uint64_t* lengths, uint64_t n, uint64_t per_bin) {
	RefBins* rb = new RefBins();
	std::vector<uint64_t> l(lengths, lengths + n);
	rb->bv = new bvec<uint8_t>(l, per_bin);
	for (uint64_t i = 0; i < n; i++) {
		DivergencePoint<uint8_t>* p = new DivergencePoint<uint8_t>(std::vector<uint8_t>(1, 1), lengths[i]);
		p->set_length(lengths[i]);
		p->set_id(i);
		rb->pts.push_back(p);
		rb->bv->insert(p);
	}
	rb->bv->insert_finalize();
	return rb;
}
void ref_bins_destroy(void* h) {
	RefBins* rb = (RefBins*)h;
	for (auto p : rb->pts) delete p;
	delete rb->bv;
	delete rb;
}
uint64_t ref_bins_count(void* h) { return ((RefBins*)h)->bv->data.size(); }
uint64_t ref_bins_layout(void* h, uint32_t* ids_out, uint64_t* sizes_out) {
	auto& data = ((RefBins*)h)->bv->data;
	uint64_t n = 0;
	for (size_t i = 0; i < data.size(); i++) {
		sizes_out[i] = data[i].size();
		for (auto& kv : data[i]) ids_out[n++] = (uint32_t)kv.first->get_id();
	}
	return n;
}
void ref_bins_range(void* h, uint64_t begin_len, uint64_t end_len, uint64_t out[5]) {
	auto r = ((RefBins*)h)->bv->get_range(begin_len, end_len);
	out[0] = r.first.first; out[1] = r.first.second; out[2] = r.second.first; out[3] = r.second.second; out[4] = r.second.is_empty ? 1 : 0;
}
// the loop of Trainer::get_close (cluster/Trainer.cpp:41-48) as OpenMP runs it: (iend - istart) iterations of istart + n
int64_t ref_bins_window(void* h, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out, uint64_t cap) {
	bvec<uint8_t>* bv = ((RefBins*)h)->bv;
	auto r = bv->get_range(begin_len, end_len);
	bvec_iterator<uint8_t> istart = bv->iter(r.first), iend = bv->iter(r.second);
	const int64_t trips = iend - istart;
	for (int64_t n = 0; n < trips; n++) {
		if ((uint64_t)n < cap) ids_out[n] = (uint32_t)(*istart).first->get_id();
		if (n + 1 < trips) ++istart;
	}
	return trips;
}
void ref_bins_mark(void* h, uint64_t bin, uint64_t at) { ((RefBins*)h)->bv->data.at(bin).at(at).second = true; }
uint64_t ref_bins_take_marked(void* h, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out) {
	bvec<uint8_t>* bv = ((RefBins*)h)->bv;
	auto r = bv->get_range(begin_len, end_len);
	std::vector<Point<uint8_t>*> out;
	bv->remove_available(r.first, r.second, out);
	for (size_t i = 0; i < out.size(); i++) ids_out[i] = (uint32_t)out[i]->get_id();
	return out.size();
}
int64_t ref_bins_take_first(void* h) {
	Point<uint8_t>* p = ((RefBins*)h)->bv->pop();
	return p ? (int64_t)p->get_id() : -1;
}
void ref_bins_erase(void* h, uint64_t bin, uint64_t at) { ((RefBins*)h)->bv->erase(bin, at); }

// Matrix::gaussJordanInverse (predict/Matrix.cpp:109-207) on an n x n row-major matrix
void ref_host_inverse(uint64_t n, const double* a, double* out) {
	matrix::Matrix m((int)n, (int)n);
	for (uint64_t i = 0; i < n; i++) for (uint64_t j = 0; j < n; j++) m.set((int)i, (int)j, a[i * n + j]);
	std::streambuf* keep = std::cout.rdbuf(nullptr);          // "Inverse does not exist" goes to stdout
	matrix::Matrix r = m.gaussJordanInverse();
	std::cout.rdbuf(keep);
	for (uint64_t i = 0; i < n; i++) for (uint64_t j = 0; j < n; j++) out[i * n + j] = r.get((int)i, (int)j);
}

long ref_train_regr(int dtype, void** first, void** second, const double* val, int n_train, int n_test, int k, uint64_t feat_flags, int max_feat, char* out, long cap,
                    double* err_out) {
	try {
		DISPATCH(dtype, return train_regr<uint8_t>(first, second, val, n_train, n_test, k, feat_flags, max_feat, out, cap, err_out),
		         return train_regr<uint16_t>(first, second, val, n_train, n_test, k, feat_flags, max_feat, out, cap, err_out),
		         return train_regr<uint32_t>(first, second, val, n_train, n_test, k, feat_flags, max_feat, out, cap, err_out),
		         return train_regr<uint64_t>(first, second, val, n_train, n_test, k, feat_flags, max_feat, out, cap, err_out));
	} catch (...) { return -2; }
	return -1;
}

long ref_train_class(int dtype, void** first, void** second, const double* val, int n_train, int n_test, int k, uint64_t feat_flags, int min_feat,
                     int max_feat, double id, char* out, long cap, double* acc_out) {
	try {
		DISPATCH(dtype, return train_class<uint8_t>(first, second, val, n_train, n_test, k, feat_flags, min_feat, max_feat, id, out, cap, acc_out),
		         return train_class<uint16_t>(first, second, val, n_train, n_test, k, feat_flags, min_feat, max_feat, id, out, cap, acc_out),
		         return train_class<uint32_t>(first, second, val, n_train, n_test, k, feat_flags, min_feat, max_feat, id, out, cap, acc_out),
		         return train_class<uint64_t>(first, second, val, n_train, n_test, k, feat_flags, min_feat, max_feat, id, out, cap, acc_out));
	} catch (...) { return -2; }
	return -2;
}

// Timed loop for bench.py's cpu_baseline (kind "reference"): all (i<j) pairs of n points through
// Feature::compute + classify, under the reference's own OpenMP schedule (cluster/Trainer.cpp:41).
double ref_time_pairs(int dtype, void* m, void** pts, int n, int reps, double* checksum) {
	if (dtype != 32) return -1;
	RefModel<uint32_t>* mod = (RefModel<uint32_t>*)m;
	Trainer<uint32_t>* t = make_trainer<uint32_t>(mod, 0.0001);
	double acc = 0;
	double t0 = omp_get_wtime();
	for (int r = 0; r < reps; r++) {
		for (int q = 0; q < n; q++) {
#pragma omp parallel for reduction(+:acc)
			for (int c = 0; c < n; c++) {
				acc += t->classify((Point<uint32_t>*)pts[c], (Point<uint32_t>*)pts[q]);
			}
		}
	}
	double dt = omp_get_wtime() - t0;
	*checksum = acc;
	delete t;
	return dt;
}

}  // extern "C"
