// msc_groups.h -- the per-group terms of the two statistics over 4-bin groups, shared by the list form (sparse.hip) and the dense
// form for histograms too small for the sparse layout (pair_features.hip).
//
// Feature<T>::markov (-> d_markov -> sim_mm) and rre_k_r (predict/Feature.cpp:1367-1393,1429-1455,1029-1062) are sums over the
// groups of four neighbouring bins that share a (k-1)-mer prefix. A group in which both histograms hold only pseudocounts
// contributes exactly 0 to either (every factor (count - 1) vanishes; both conditional distributions are uniform, log 1 = 0), so
// callers may skip such groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// markov += sum_j (q_j - 1)(log p_j - log psum) + (p_j - 1)(log q_j - log qsum)       (the reference halves the total at the end)
// rre    += sum_j p_j log(cp_j / avg_j) / psum, then sum_j q_j log(cq_j / avg_j) / qsum  with cp = p / psum, cq = q / qsum
__device__ __forceinline__ void msc_group_terms(const uint32_t (&p)[4], const uint32_t (&q)[4], double& markov, double& rre) {
	const double sp = (double)((uint64_t)p[0] + p[1] + p[2] + p[3]);
	const double sq = (double)((uint64_t)q[0] + q[1] + q[2] + q[3]);
	const double lsp = log(sp), lsq = log(sq);
	double ip = 0.0, iq = 0.0;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const double pj = (double)p[j], qj = (double)q[j];
		markov += (double)(q[j] - 1u) * (log(pj) - lsp);
		markov += (double)(p[j] - 1u) * (log(qj) - lsq);
		const double cp = pj / sp, cq = qj / sq;
		const double avg = 0.5 * (cp + cq);
		ip += pj * log(cp / avg) / sp;
		iq += qj * log(cq / avg) / sq;
	}
	rre += ip;
	rre += iq;
}

// markov(a, a) of one histogram, one group: sum_j (a_j - 1)(log a_j - log group sum) (its two equal terms per bin are halved later)
__device__ __forceinline__ void msc_group_self(const uint32_t (&v)[4], double& total) {
	const double ls = log((double)((uint64_t)v[0] + v[1] + v[2] + v[3]));
#pragma unroll
	for (int j = 0; j < 4; j++) total += (double)(v[j] - 1u) * (log((double)v[j]) - ls);
}
