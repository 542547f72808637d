// msc_train.hip -- SURVEY.md 8(f2): model training on caller-supplied labelled pairs (host logic over the GPU feature table).
//
// Replaces the feature-selection half of Predictor<T>::train (predict/Predictor.cpp:876-975): the raw statistics of every
// training/testing pair come from the streaming kernels (msc_pair_features_raw, one 1 x M pass per distinct second point) instead
// of one CPU pass per statistic per pair (calculate_table, predict/BestFirstSelector.cpp:113-128); what remains is small host
// work restated here in the reference's evaluation order so that the SAME model comes out:
//   - min-max normalisation over the TRAINING pairs (calculate_table rebuilds the single features, so the training+testing
//     normalisation of Predictor::train :889-894 is discarded), polarity per feat_is_sim (predict/Feature.cpp:137-154,216-268);
//   - the candidate list of Predictor::add_feats (:201-220): every single and every pair of singles as xy / x2y2 (/ x2y / xy2);
//   - best-first search over sets of candidates (predict/BestFirstSelector.cpp:28-53,146-250): a child toggles one candidate,
//     children are scored by TEST accuracy of a least-squares fit on +-1 labels, the open list is a max-heap on accuracy;
//   - GLM::train = normal equations through Matrix::pseudoInverse / gaussJordanInverse (predict/GLM.cpp:20-23,
//     predict/Matrix.cpp:109-221), including its "not invertible -> hand back the input" behaviour;
//   - the class block of the weights file as Predictor::save / write_to print it (:28-44,82-121).
// Data generation (mutated templates with known identity, predict/Predictor.cpp:519-710) stays out of scope: pairs and labels are inputs.
// std::set / std::priority_queue order the search exactly as the reference's own containers do (same libstdc++).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <queue>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "msc_internal.h"
#include "../host/msc_hostmath.hpp"

int msc_feat_is_sim(uint64_t f);        // msc_api.hip
int msc_set_error(msc_ctx* ctx, int code, const char* msg);
bool msc_ctx_owns(const msc_ctx* ctx, const msc_hist_set* set);

namespace {

// enum class Combo { xy, x2y2, xy2, x2y } -- this ORDER is the reference's (predict/Feature.h:66-71): sets of candidates sort by it
enum Cmb { C_XY = 0, C_X2Y2 = 1, C_XY2 = 2, C_X2Y = 3 };
typedef std::pair<uint64_t, int> Cand;          // (flags of one or two singles, Cmb)
typedef std::set<Cand> CandSet;

typedef msc::hostmath::Matrix Mat;      // the reference's matrix::Matrix, restated in host/msc_hostmath.hpp (product, transpose, Gauss-Jordan)
using msc::hostmath::product;
using msc::hostmath::pseudo_inverse;
using msc::hostmath::transposed;

struct Table {
	std::vector<uint64_t> singles;            // ascending bit order == Feature::lookup after calculate_table
	std::vector<double> mins, maxs;
	std::vector<int> is_sim;
	std::vector<std::vector<double> > norm;   // [pair][single]: normalised, polarity applied
	std::vector<double> label;                // +1 / -1
	size_t n_train = 0, n_test = 0;
};

int index_of(const Table& t, uint64_t f) { return (int)(std::find(t.singles.begin(), t.singles.end(), f) - t.singles.begin()); }

// Feature::operator() (predict/Feature.h:205-239); singles of a candidate in ascending bit order
double combo_value(const Table& t, size_t pair, const Cand& c) {
	int idx[2], n = 0;
	for (uint64_t f = 1; f <= c.first; f <<= 1) if (c.first & f) idx[n++] = index_of(t, f);
	const std::vector<double>& v = t.norm[pair];
	switch (c.second) {
	case C_XY: { double p = 1; for (int i = 0; i < n; i++) p *= v[idx[i]]; return p; }
	case C_X2Y2: { double p = 1; for (int i = 0; i < n; i++) p *= v[idx[i]] * v[idx[i]]; return p; }
	case C_XY2: return v[idx[0]] * v[idx[1]] * v[idx[1]];
	default: return v[idx[0]] * v[idx[0]] * v[idx[1]];
	}
}

// generate_feat_mat (predict/FeatureSelector.cpp:10-38): [1, combo 0, combo 1, ...] per pair, combos in set order
Mat feature_matrix(const Table& t, const CandSet& set, size_t first, size_t count) {
	Mat x(count, set.size() + 1);
	for (size_t r = 0; r < count; r++) {
		x.at(r, 0) = 1;
		size_t c = 1;
		for (const Cand& cd : set) x.at(r, c++) = combo_value(t, first + r, cd);
	}
	return x;
}

// GLM::train (predict/GLM.cpp:20-23): weights = pinv(X^T X) * X^T * y, products taken left to right
std::vector<double> glm_train(const Mat& x, const std::vector<double>& y) {
	const Mat xt = transposed(x);
	const Mat normal = product(xt, x);
	Mat ycol(y.size(), 1);
	for (size_t i = 0; i < y.size(); i++) ycol.at(i, 0) = y[i];
	const Mat w = product(product(pseudo_inverse(normal), xt), ycol);
	return w.v;
}

// GLM::predict + accuracy (predict/GLM.cpp:30-66, FeatureSelector::class_test :93-103): round(logistic(Xw)), 0 counted as -1
double accuracy(const Mat& x, const std::vector<double>& w, const std::vector<double>& label, size_t first) {
	size_t same = 0;
	for (size_t r = 0; r < x.rows; r++) {
		double s = 0;
		for (size_t k = 0; k < w.size(); k++) s = std::fma(x.at(r, k), w[k], s);      // features * weights, Matrix::operator* as built (msc_hostmath.hpp)
		double p = round(1.0 / (1 + exp(-s)));
		if (p == 0) p = -1;
		if (p == label[first + r]) same++;
	}
	return ((double)same * 100) / (double)x.rows;
}

// feature_accuracy (predict/BestFirstSelector.cpp:129-143): fit on the training pairs, score on the testing pairs
double set_accuracy(const Table& t, const CandSet& set) {
	const Mat xtr = feature_matrix(t, set, 0, t.n_train);
	const std::vector<double> ytr(t.label.begin(), t.label.begin() + (long)t.n_train);
	const std::vector<double> w = glm_train(xtr, ytr);
	return accuracy(feature_matrix(t, set, t.n_train, t.n_test), w, t.label, t.n_train);
}

struct HeapLess {
	bool operator()(const std::pair<CandSet, double>& a, const std::pair<CandSet, double>& b) const { return a.second < b.second; }
};
typedef std::priority_queue<std::pair<CandSet, double>, std::vector<std::pair<CandSet, double> >, HeapLess> Heap;

// children_of (:30-53): toggle each candidate in turn; keep what is non-empty and neither closed nor open
std::vector<CandSet> children_of(const CandSet& cur, const std::vector<Cand>& all, const std::set<CandSet>& closed, const std::set<CandSet>& open) {
	std::vector<CandSet> out;
	for (const Cand& c : all) {
		CandSet tmp = cur;
		if (!tmp.erase(c)) tmp.insert(c);
		if (!tmp.empty() && !closed.count(tmp) && !open.count(tmp)) out.push_back(tmp);
	}
	return out;
}

void evaluate(const Table& t, const std::vector<CandSet>& items, std::set<CandSet>& open, Heap& heap) {
	for (const CandSet& it : items) {          // the reference's loop is an omp parallel for; this is its one-thread order
		const double acc = set_accuracy(t, it);
		open.insert(it);
		heap.push(std::make_pair(it, acc));
	}
}

std::string fmt15(double v) {
	char b[64];
	snprintf(b, sizeof b, "%.15g", v);      // out << std::setprecision(digits10) << v
	return b;
}


// The feature table both trainings start from: raw statistics of every pair through the streaming kernels, min-max normalisation over
// the TRAINING pairs with the reference's polarity (see the header), labels left to the caller. who = the entry point's name.
int build_table(msc_ctx* ctx, const msc_hist_set* pts, const uint32_t* first_slots, const uint32_t* second_slots, uint64_t n_train, uint64_t n_test, uint64_t feat_flags,
                Table& t, const char* who) {
	char msg[256];
	const size_t n = (size_t)(n_train + n_test);
	t.n_train = (size_t)n_train;
	t.n_test = (size_t)n_test;
	for (uint64_t f = 1; f <= feat_flags; f <<= 1) if (feat_flags & f) { t.singles.push_back(f); t.is_sim.push_back(msc_feat_is_sim(f)); }
	const size_t ns = t.singles.size();
	// ---- raw statistics of every pair: func(*pair.first, *pair.second), one streaming pass per distinct second point
	std::vector<std::vector<double> > raw(n, std::vector<double>(ns));
	{
		std::map<uint32_t, std::vector<size_t> > by_second;
		for (size_t i = 0; i < n; i++) by_second[second_slots[i]].push_back(i);
		std::vector<uint32_t> cands;
		std::vector<double> out;
		for (const auto& kv : by_second) {
			cands.clear();
			for (size_t i : kv.second) cands.push_back(first_slots[i]);
			out.assign(cands.size() * ns, 0.0);
			const int r = msc_pair_features_raw(ctx, pts, cands.data(), cands.size(), pts, kv.first, MSC_ORDER_CAND_FIRST, feat_flags, out.data());
			if (r) return r;
			for (size_t j = 0; j < kv.second.size(); j++) for (size_t s = 0; s < ns; s++) raw[kv.second[j]][s] = out[j * ns + s];
		}
	}
	// ---- Feature::normalize over the training pairs (predict/Feature.cpp:216-268); the odd start values are the reference's
	t.mins.assign(ns, DBL_MAX);
	t.maxs.assign(ns, DBL_MIN);
	for (size_t s = 0; s < ns; s++) {
		for (size_t i = 0; i < t.n_train; i++) {
			if (raw[i][s] < t.mins[s]) t.mins[s] = raw[i][s];
			if (raw[i][s] > t.maxs[s]) t.maxs[s] = raw[i][s];
		}
		if (fabs(t.maxs[s] - t.mins[s]) <= 0.000000001 || std::isinf(t.maxs[s]) || std::isinf(t.mins[s])) {      // the reference throws
			snprintf(msg, sizeof msg, "%s: a statistic is constant (or infinite) over the training pairs (Feature::normalize throws)", who);
			return msc_set_error(ctx, MSC_ERR_NAN, msg);
		}
	}
	t.norm.assign(n, std::vector<double>(ns));
	t.label.resize(n);
	for (size_t i = 0; i < n; i++) {
		for (size_t s = 0; s < ns; s++) {
			const double v = (raw[i][s] - t.mins[s]) / (t.maxs[s] - t.mins[s]);      // normalize_cache, :137-154
			if (std::isnan(v)) { snprintf(msg, sizeof msg, "%s: NaN after normalisation", who); return msc_set_error(ctx, MSC_ERR_NAN, msg); }
			t.norm[i][s] = t.is_sim[s] ? v : 1 - v;
		}
	}
	return MSC_OK;
}

// Predictor::add_feats (:201-220)
std::vector<Cand> candidates_of(uint64_t feat_flags) {
	std::vector<Cand> all;
	for (uint64_t i = 1; i <= feat_flags; i <<= 1) {
		if (!(i & feat_flags)) continue;
		for (uint64_t j = 1; j <= i; j <<= 1) {
			if (!(j & feat_flags)) continue;
			all.push_back(Cand(i | j, C_XY));
			all.push_back(Cand(i | j, C_X2Y2));
			if (i != j) { all.push_back(Cand(i | j, C_X2Y)); all.push_back(Cand(i | j, C_XY2)); }
		}
	}
	return all;
}

// [1, combo 0, combo 1, ...] per pair for combos in a GIVEN order (Feature::combos of the greedy search: order of acceptance)
Mat feature_matrix(const Table& t, const std::vector<Cand>& list, size_t first, size_t count) {
	Mat x(count, list.size() + 1);
	for (size_t r = 0; r < count; r++) {
		x.at(r, 0) = 1;
		for (size_t c = 0; c < list.size(); c++) x.at(r, c + 1) = combo_value(t, first + r, list[c]);
	}
	return x;
}

// FeatureSelector::regression_train / regression_test (predict/FeatureSelector.cpp:41-55,77-89): mean |X w - y|
double mean_abs_error(const Mat& x, const std::vector<double>& w, const std::vector<double>& y, size_t first) {
	double sum = 0;
	for (size_t r = 0; r < x.rows; r++) {
		double s = 0;
		for (size_t k = 0; k < w.size(); k++) s = std::fma(x.at(r, k), w[k], s);      // Matrix::operator* as built (msc_hostmath.hpp)
		sum += fabs(s - y[first + r]);
	}
	return sum / (double)x.rows;
}

// the block of a weights file (Predictor::write_to, :82-121): intercept, combos in the given order, singles in order of first appearance
std::string block_text(const Table& t, const std::vector<Cand>& combos, const std::vector<double>& w) {
	std::vector<uint64_t> lookup;
	for (const Cand& c : combos)
		for (uint64_t f = 1; f <= c.first; f <<= 1)
			if ((c.first & f) && std::find(lookup.begin(), lookup.end(), f) == lookup.end()) lookup.push_back(f);
	static const int file_code[4] = {0, 3, 1, 2};      // xy 0, xy2 1, x2y 2, x2y2 3 in the file
	std::string text = "\nn_combos: " + std::to_string(combos.size()) + "\n" + fmt15(w[0]) + "\n";
	size_t col = 1;
	for (const Cand& c : combos) text += std::to_string(file_code[c.second]) + " " + std::to_string((unsigned long long)c.first) + " " + fmt15(w[col++]) + "\n";
	text += "\nn_singles: " + std::to_string(lookup.size()) + "\n";
	for (uint64_t f : lookup) {
		const int i = index_of(t, f);
		text += std::to_string((unsigned long long)f) + " " + fmt15(t.mins[(size_t)i]) + " " + fmt15(t.maxs[(size_t)i]) + "\n";
	}
	return text;
}

}  // namespace

extern "C" int msc_train_class(msc_ctx* ctx, const msc_hist_set* pts, const uint32_t* first_slots, const uint32_t* second_slots, const double* vals,
                               uint64_t n_train, uint64_t n_test, uint64_t feat_flags, int min_feat, int max_feat, double id, char* text_out, size_t cap,
                               double* train_acc, double* test_acc) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (!pts || !first_slots || !second_slots || !vals || !text_out) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: NULL argument");
	if (!msc_ctx_owns(ctx, pts)) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: the histogram set belongs to another context");
	if (n_train == 0 || n_test == 0) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: needs training and testing pairs");
	if (min_feat < 1 || max_feat < min_feat) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: need 1 <= min_feat <= max_feat");
	// a model msc_model_create could not load afterwards is refused before any work is done
	if (max_feat > MSC_MAX_COMBOS) return msc_set_error(ctx, MSC_ERR_UNSUPPORTED, "msc_train_class: max_feat exceeds MSC_MAX_COMBOS (8)");
	if (feat_flags == 0 || (feat_flags & ~(uint64_t)MSC_FEAT_SLOW))
		return msc_set_error(ctx, MSC_ERR_UNSUPPORTED, "msc_train_class: feat_flags must be a non-empty subset of MSC_FEAT_SLOW");
	const size_t n = (size_t)(n_train + n_test);
	Table t;
	{
		const int r = build_table(ctx, pts, first_slots, second_slots, n_train, n_test, feat_flags, t, "msc_train_class");
		if (r) return r;
	}
	for (size_t i = 0; i < n; i++) t.label[i] = vals[i] >= id ? 1 : -1;      // generate_feat_mat, predict/FeatureSelector.cpp:26-28
	const std::vector<Cand> all = candidates_of(feat_flags);
	// ---- BestFirstSelector::train_class (predict/BestFirstSelector.cpp:187-250): best-first search over candidate sets. A set is
	// expanded by toggling one candidate at a time; expansion order = highest testing accuracy first (max-heap, ties as
	// std::priority_queue leaves them). The search ends when the open list holds a set larger than max_feat, or when the best
	// admissible set (min_feat..max_feat candidates, strictly better accuracy than any before, starting from -100) has not changed
	// for three expansions while the open list already holds sets larger than min_feat.
	CandSet best;
	std::set<CandSet> closed, open;
	Heap heap;
	double best_acc = -100;
	long expansions = 0, expansion_of_best = 0;
	evaluate(t, children_of(CandSet(), all, closed, open), open, heap);
	while (!open.empty()) {
		size_t largest_open = 0;
		for (const CandSet& s : open) largest_open = std::max(largest_open, s.size());
		const bool stalled = expansions - expansion_of_best >= 3 && (long)largest_open > min_feat;
		if ((long)largest_open > max_feat || stalled) break;
		const CandSet cur = heap.top().first;
		const double acc = heap.top().second;
		heap.pop();
		open.erase(cur);
		closed.insert(cur);
		const long size = (long)cur.size();
		if (acc > best_acc && size >= min_feat && size <= max_feat) { best = cur; best_acc = acc; expansion_of_best = expansions; }
		evaluate(t, children_of(cur, all, closed, open), open, heap);
		expansions++;
	}
	if (best.empty()) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: the search found no model with min_feat..max_feat combos");
	const Mat xtr = feature_matrix(t, best, 0, t.n_train);
	const std::vector<double> ytr(t.label.begin(), t.label.begin() + (long)t.n_train);
	const std::vector<double> w = glm_train(xtr, ytr);
	if (train_acc) *train_acc = accuracy(xtr, w, t.label, 0);
	if (test_acc) *test_acc = accuracy(feature_matrix(t, best, t.n_train, t.n_test), w, t.label, t.n_train);
	// ---- Predictor::save + write_to (:28-44,82-121); singles in order of first appearance over the chosen candidates (load_feat, :77-111)
	char head[256];
	snprintf(head, sizeof head, "k: %d\nmode: 1\nmax_features: %d\nID: %g\nDatatype: uint%d_t\nfeature_set: %llu\n", msc_hist_set_k(pts), max_feat, id,
	         msc_hist_set_dtype(pts), (unsigned long long)feat_flags);
	std::string text = head;
	text += block_text(t, std::vector<Cand>(best.begin(), best.end()), w);
	if (text.size() + 1 > cap) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_class: text_out is too small for the weights file");
	memcpy(text_out, text.c_str(), text.size() + 1);
	return MSC_OK;
}

// Predictor<T>::train_regr (predict/Predictor.cpp:977-985) -> GreedySelector<T>::train_regression (predict/GreedySelector.cpp:11-76) on
// labelled pairs: the regression model fastcar's work() reads identities from (Predictor::similarity -> p_predict, :284-300). Greedy
// forward selection: up to max_feat rounds; a round fits, for every candidate not yet used, a least-squares model of the IDENTITY VALUES
// on [1, accepted combos ..., candidate] over the training pairs and takes the candidate with the smallest mean absolute error over the
// testing pairs -- kept only if that beats every earlier round (start: 1 000 000). Combos stay in order of acceptance. The reference's own
// function cannot be run to its end (it has no return statement; `fastcar --dump` dies there): oracle/ref_harness.cpp follows its
// body on the reference's objects, and the fixture tests/golden/train_regr_*.json holds what that gives.
// text_out: a complete weights file with `mode: 2` (regression block only); msc_model_parse(text, 1) reads it.
extern "C" int msc_train_regr(msc_ctx* ctx, const msc_hist_set* pts, const uint32_t* first_slots, const uint32_t* second_slots, const double* vals, uint64_t n_train,
                              uint64_t n_test, uint64_t feat_flags, int max_feat, double id, char* text_out, size_t cap, double* train_err, double* test_err) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (!pts || !first_slots || !second_slots || !vals || !text_out) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: NULL argument");
	if (!msc_ctx_owns(ctx, pts)) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: the histogram set belongs to another context");
	if (n_train == 0 || n_test == 0) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: needs training and testing pairs");
	if (max_feat < 1) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: need max_feat >= 1");
	if (max_feat > MSC_MAX_COMBOS) return msc_set_error(ctx, MSC_ERR_UNSUPPORTED, "msc_train_regr: max_feat exceeds MSC_MAX_COMBOS (8)");
	if (feat_flags == 0 || (feat_flags & ~(uint64_t)MSC_FEAT_SLOW))
		return msc_set_error(ctx, MSC_ERR_UNSUPPORTED, "msc_train_regr: feat_flags must be a non-empty subset of MSC_FEAT_SLOW");
	const size_t n = (size_t)(n_train + n_test);
	Table t;
	{
		const int r = build_table(ctx, pts, first_slots, second_slots, n_train, n_test, feat_flags, t, "msc_train_regr");
		if (r) return r;
	}
	for (size_t i = 0; i < n; i++) t.label[i] = vals[i];      // generate_feat_mat with cutoff < 0: the value itself (predict/FeatureSelector.cpp:29-31)
	const std::vector<Cand> all = candidates_of(feat_flags);
	const std::vector<double> ytr(t.label.begin(), t.label.begin() + (long)t.n_train);
	std::vector<Cand> chosen;
	std::vector<size_t> used;
	double abs_best = 1000000;
	for (int round = 1; round <= max_feat; round++) {
		double best_err = abs_best;
		size_t best_idx = (size_t)-1;
		for (size_t i = 0; i < all.size(); i++) {
			if (std::find(used.begin(), used.end(), i) != used.end()) continue;
			std::vector<Cand> trial = chosen;
			trial.push_back(all[i]);
			const std::vector<double> w = glm_train(feature_matrix(t, trial, 0, t.n_train), ytr);
			const double err = mean_abs_error(feature_matrix(t, trial, t.n_train, t.n_test), w, t.label, t.n_train);
			if (err < best_err) { best_err = err; best_idx = i; }
		}
		if (best_err < abs_best) { chosen.push_back(all[best_idx]); abs_best = best_err; used.push_back(best_idx); }
	}
	if (chosen.empty()) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: no candidate reaches a mean error below the reference's starting value");
	const Mat xtr = feature_matrix(t, chosen, 0, t.n_train);
	const std::vector<double> w = glm_train(xtr, ytr);
	if (train_err) *train_err = mean_abs_error(xtr, w, t.label, 0);
	if (test_err) *test_err = mean_abs_error(feature_matrix(t, chosen, t.n_train, t.n_test), w, t.label, t.n_train);
	char head[256];
	snprintf(head, sizeof head, "k: %d\nmode: 2\nmax_features: %d\nID: %g\nDatatype: uint%d_t\nfeature_set: %llu\n", msc_hist_set_k(pts), max_feat, id,
	         msc_hist_set_dtype(pts), (unsigned long long)feat_flags);
	const std::string text = std::string(head) + block_text(t, chosen, w);
	if (text.size() + 1 > cap) return msc_set_error(ctx, MSC_ERR_INVALID_ARG, "msc_train_regr: text_out is too small for the weights file");
	memcpy(text_out, text.c_str(), text.size() + 1);
	return MSC_OK;
}
