// msc_hostmath.hpp -- the small dense linear algebra behind GLM::train; plain C++, no device code.
//
// The reference fits its linear model through the normal equations (predict/GLM.cpp:20-23): weights = pinv(X^T X) X^T y with
// Matrix::pseudoInverse / Matrix::gaussJordanInverse (predict/Matrix.cpp:109-221) on matrices of at most 5 x 5. Which model the
// best-first search ends on depends on the accuracies of those fits, so the arithmetic has to ROUND like the reference's: the same
// operations on the same operands (the order of independent element operations is free). What is reproduced, as behaviour:
//   - elimination without pivoting for size: a pivot is used as it stands unless it is exactly 0, in which case the first row
//     below with a non-zero entry in that column is swapped in (none: the matrix counts as singular);
//   - forward pass column by column (normalise the pivot row unless the pivot is exactly 1, then clear the column below),
//     backward pass from the last column up (clear the column above);
//   - every update  x - factor * y  and every step  s + a * b  of a matrix product is ONE fused multiply-add: the reference is
//     built with -O3 -march=native (CMakeLists.txt:96-99) under GCC's default -ffp-contract=fast, so on any FMA-capable x86 those
//     expressions are single-rounded (oracle/_ref, built for x86-64-v3, is such a build); std::fma states that explicitly and
//     independently of this file's own compiler flags. Every normalisation is a true division;
//   - the reduced matrix must be EXACTLY the identity; if it is not, or no pivot row exists, the ORIGINAL matrix is handed
//     back as the "inverse" (the reference prints "Inverse does not exist" and carries on with it).
// tests/test_driver_cpu.py holds inverse() to Matrix::gaussJordanInverse of the compiled reference on random matrices, bit for bit.
#pragma once
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace msc {
namespace hostmath {

struct Matrix {                       // row major
	size_t rows = 0, cols = 0;
	std::vector<double> v;
	Matrix() {}
	Matrix(size_t r, size_t c) : rows(r), cols(c), v(r * c, 0.0) {}
	double& at(size_t r, size_t c) { return v[r * cols + c]; }
	double at(size_t r, size_t c) const { return v[r * cols + c]; }
};

// Matrix::operator* (predict/Matrix.cpp:76-96): every entry is a left-to-right sum of products
inline Matrix product(const Matrix& a, const Matrix& b) {
	Matrix out(a.rows, b.cols);
	for (size_t i = 0; i < a.rows; i++)
		for (size_t j = 0; j < b.cols; j++) {
			double s = 0;
			for (size_t k = 0; k < a.cols; k++) s = std::fma(a.at(i, k), b.at(k, j), s);
			out.at(i, j) = s;
		}
	return out;
}

inline Matrix transposed(const Matrix& a) {
	Matrix t(a.cols, a.rows);
	for (size_t i = 0; i < a.rows; i++) for (size_t j = 0; j < a.cols; j++) t.at(j, i) = a.at(i, j);
	return t;
}

// Gauss-Jordan on the augmented matrix [A | I]: one row operation touches both halves, so they live in one 2n-wide row.
inline Matrix inverse(const Matrix& original) {
	const size_t n = original.rows, w = 2 * n;
	std::vector<double> aug(n * w, 0.0);
	for (size_t r = 0; r < n; r++) {
		for (size_t c = 0; c < n; c++) aug[r * w + c] = original.at(r, c);
		aug[r * w + n + r] = 1.0;
	}
	double* const m = aug.data();
	const auto divide_row = [&](size_t r, double by) { for (size_t c = 0; c < w; c++) m[r * w + c] = m[r * w + c] / by; };
	const auto subtract_multiple = [&](size_t r, double factor, size_t of) {          // row r -= factor * row `of`
		for (size_t c = 0; c < w; c++) m[r * w + c] = std::fma(-factor, m[of * w + c], m[r * w + c]);
	};
	for (size_t col = 0; col < n; col++) {                                           // forward: unit pivot, zeros below it
		if (m[col * w + col] == 0) {
			size_t other = col + 1;
			while (other < n && m[other * w + col] == 0) other++;
			if (other == n) return original;                                          // no pivot: "singular"
			for (size_t c = 0; c < w; c++) std::swap(m[col * w + c], m[other * w + c]);
		}
		if (m[col * w + col] != 1) divide_row(col, m[col * w + col]);
		for (size_t r = col + 1; r < n; r++) if (m[r * w + col] != 0) subtract_multiple(r, m[r * w + col], col);
	}
	for (size_t col = n; col-- > 0;)                                                  // backward: zeros above every pivot
		for (size_t r = 0; r < col; r++) if (m[r * w + col] != 0) subtract_multiple(r, m[r * w + col], col);
	Matrix inv(n, n);
	for (size_t r = 0; r < n; r++)
		for (size_t c = 0; c < n; c++) {
			if (m[r * w + c] != (r == c ? 1.0 : 0.0)) return original;                // the left half must be the identity, exactly
			inv.at(r, c) = m[r * w + n + c];
		}
	return inv;
}

// Matrix::pseudoInverse (:209-221) for rows >= cols: (A^T A)^-1 A^T
inline Matrix pseudo_inverse(const Matrix& a) {
	const Matrix t = transposed(a);
	return product(inverse(product(t, a)), t);
}

}  // namespace hostmath
}  // namespace msc
