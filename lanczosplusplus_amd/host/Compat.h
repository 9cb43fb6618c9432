// Compat.h -- the small part of the PsimagLite surface the stored-CSR path needs,
// written from scratch because PsimagLite is not vendored with the reference and absent here:
// an InputNg-style reader (Label=value / "Label n v1..vn"), chain/ladder connection matrices,
// a CSR container with 64-bit row pointers and the SparseRow accumulator used by setupHamiltonian.
// It is host plumbing around the engine, not part of it.
#ifndef LPP_HOST_COMPAT_H
#define LPP_HOST_COMPAT_H

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

typedef std::size_t SizeType;

namespace LppHost {

typedef std::string String;

struct RuntimeError : public std::runtime_error {
	explicit RuntimeError(const String& s) : std::runtime_error(s) { }
};

inline void err(const String& s) { throw RuntimeError(s); }

template <typename T> struct Real {
	typedef T Type;
};
template <typename T> struct Real<std::complex<T>> {
	typedef T Type;
};
inline double real(double x) { return x; }
inline double imag(double) { return 0.0; }
inline double real(const std::complex<double>& x) { return x.real(); }
inline double imag(const std::complex<double>& x) { return x.imag(); }
inline double conj(double x) { return x; }
inline std::complex<double> conj(const std::complex<double>& x) { return std::conj(x); }

// ---------------------------------------------------------------------------------------------
// Input: scalars `Label=value`, vectors `Label n v1 ... vn`; repeated labels are consumed in order
// (one `Connectors` entry per Hamiltonian term), like InputNg's readline/read.
// ---------------------------------------------------------------------------------------------
class InputReadable {
public:
	explicit InputReadable(const String& filename) : filename_(filename)
	{
		std::ifstream fin(filename.c_str());
		if (!fin) err("InputReadable: cannot open " + filename + "\n");
		std::vector<String> toks;
		String t;
		while (fin >> t) toks.push_back(t);
		for (SizeType i = 0; i < toks.size();) {
			const String& s = toks[i];
			const SizeType eq = s.find('=');
			if (eq != String::npos) {
				scalars_[s.substr(0, eq + 1)].push_back(s.substr(eq + 1));
				i++;
			} else if (i + 1 < toks.size() && isInteger(toks[i + 1]) && !isNumber(s)) {
				const SizeType n = std::stoul(toks[i + 1]);
				std::vector<double> v;
				for (SizeType k = 0; k < n && i + 2 + k < toks.size(); k++) v.push_back(std::stod(toks[i + 2 + k]));
				if (v.size() != n) err("InputReadable: truncated vector " + s + "\n");
				vectors_[s].push_back(v);
				i += 2 + n;
			} else {
				i++;
			}
		}
	}

	const String& filename() const { return filename_; }

	// label includes the trailing '=' as in the reference's call sites (io.readline(x,"Model="))
	template <typename T> void readline(T& x, const String& label)
	{
		auto it = scalars_.find(label);
		if (it == scalars_.end()) throw RuntimeError("InputReadable: label " + label + " not found\n");
		SizeType& pos = cursorS_[label];
		const SizeType use = std::min(pos, it->second.size() - 1);
		std::istringstream iss(it->second[use]);
		iss >> x;
		if (pos + 1 < it->second.size()) pos++;
	}

	template <typename T> void read(std::vector<T>& v, const String& label)
	{
		auto it = vectors_.find(label);
		if (it == vectors_.end()) throw RuntimeError("InputReadable: vector " + label + " not found\n");
		SizeType& pos = cursorV_[label];
		const SizeType use = std::min(pos, it->second.size() - 1);
		v.assign(it->second[use].begin(), it->second[use].end());
		if (pos + 1 < it->second.size()) pos++;
	}

	bool has(const String& label) const { return scalars_.count(label) > 0 || vectors_.count(label) > 0; }

private:
	static bool isInteger(const String& s)
	{
		if (s.empty()) return false;
		for (char c : s)
			if (c < '0' || c > '9') return false;
		return true;
	}
	static bool isNumber(const String& s)
	{
		std::istringstream iss(s);
		double d;
		char c;
		return (iss >> d) && !(iss >> c);
	}
	String filename_;
	std::map<String, std::vector<String>> scalars_;
	std::map<String, std::vector<std::vector<double>>> vectors_;
	std::map<String, SizeType> cursorS_, cursorV_;
};

// ---------------------------------------------------------------------------------------------
// Geometry: geometry(i, orb_i, j, orb_j, term) for GeometryKind=chain|ladder, ConstantValues.
// Conventions restated from memory of PsimagLite (UNVERIFIED): ladder site = x*leg + y, one
// `Connectors` entry for a chain term, two (along x, along y) for a ladder term.
// ---------------------------------------------------------------------------------------------
template <typename ComplexOrRealType_> class Geometry {
public:
	typedef ComplexOrRealType_ ComplexOrRealType;

	explicit Geometry(InputReadable& io)
	{
		io.readline(n_, "TotalNumberOfSites=");
		SizeType nterms = 1;
		io.readline(nterms, "NumberOfTerms=");
		int px = 0, py = 0;
		if (io.has("IsPeriodicX=")) io.readline(px, "IsPeriodicX=");
		if (io.has("IsPeriodicY=")) io.readline(py, "IsPeriodicY=");
		for (SizeType t = 0; t < nterms; t++) {
			String kind("chain");
			io.readline(kind, "GeometryKind=");
			std::vector<double> m(n_ * n_, 0.0);
			std::vector<double> c;
			if (kind == "chain") {
				io.read(c, "Connectors");
				for (SizeType i = 0; i + 1 < n_; i++) add(m, i, i + 1, c[0]);
				if (px && n_ > 2) add(m, n_ - 1, 0, c[0]);
			} else if (kind == "ladder") {
				SizeType leg = 2;
				io.readline(leg, "LadderLeg=");
				if (n_ % leg) err("Geometry: TotalNumberOfSites not a multiple of LadderLeg\n");
				const SizeType lx = n_ / leg;
				std::vector<double> cy;
				io.read(c, "Connectors");
				io.read(cy, "Connectors");
				for (SizeType x = 0; x < lx; x++)
					for (SizeType y = 0; y < leg; y++) {
						const SizeType s = x * leg + y;
						if (x + 1 < lx)
							add(m, s, (x + 1) * leg + y, c[0]);
						else if (px && lx > 2)
							add(m, s, y, c[0]);
						if (y + 1 < leg)
							add(m, s, x * leg + y + 1, cy[0]);
						else if (py && leg > 2)
							add(m, s, x * leg, cy[0]);
					}
			} else {
				err("Geometry: unsupported GeometryKind=" + kind + "\n");
			}
			terms_.push_back(m);
		}
	}

	SizeType numberOfSites() const { return n_; }
	SizeType terms() const { return terms_.size(); }
	ComplexOrRealType operator()(SizeType i, SizeType, SizeType j, SizeType, SizeType term) const
	{
		return ComplexOrRealType(terms_[term][i * n_ + j]);
	}
	const std::vector<double>& term(SizeType t) const { return terms_[t]; }

private:
	void add(std::vector<double>& m, SizeType a, SizeType b, double v)
	{
		if (a == b) return;
		m[a * n_ + b] += v;
		m[b * n_ + a] += v;
	}
	SizeType n_ = 0;
	std::vector<std::vector<double>> terms_;
};

// ---------------------------------------------------------------------------------------------
// CSR container (64-bit row pointers, 32-bit columns: the layout the engine takes)
// ---------------------------------------------------------------------------------------------
template <typename T> class CrsMatrix {
public:
	typedef T value_type;
	void resize(SizeType nrows, SizeType ncols)
	{
		nrows_ = nrows;
		ncols_ = ncols;
		rowptr_.assign(nrows + 1, 0);
		colind_.clear();
		values_.clear();
	}
	void setRow(SizeType i, SizeType n) { rowptr_[i] = (int64_t)n; }
	void pushCol(SizeType c) { colind_.push_back((int32_t)c); }
	void pushValue(const T& v) { values_.push_back(v); }
	SizeType rows() const { return nrows_; }
	SizeType cols() const { return ncols_; }
	SizeType nonZeros() const { return colind_.size(); }
	int64_t getRowPtr(SizeType i) const { return rowptr_[i]; }
	int32_t getCol(SizeType k) const { return colind_[k]; }
	const T& getValue(SizeType k) const { return values_[k]; }
	const std::vector<int64_t>& rowptr() const { return rowptr_; }
	const std::vector<int32_t>& colind() const { return colind_; }
	const std::vector<T>& values() const { return values_; }
	std::vector<int64_t>& rowptr() { return rowptr_; }
	std::vector<int32_t>& colind() { return colind_; }
	std::vector<T>& values() { return values_; }

private:
	SizeType nrows_ = 0, ncols_ = 0;
	std::vector<int64_t> rowptr_;
	std::vector<int32_t> colind_;
	std::vector<T> values_;
};

// ---------------------------------------------------------------------------------------------
// Dense matrix + Hermitian eigen-solver: what DefaultSymmetry::fullDiag needs (DefaultSymmetry.h:80-93: toDense + diag(fm,eigs,'V'),
// LAPACK behind PsimagLite there).  Householder tridiagonalisation + implicit QL with accumulated transformations;
// complex Hermitian matrices go through the real embedding [[A,-B],[B,A]].  Host-only, for the <= 4900-row fallback.
// ---------------------------------------------------------------------------------------------
template <typename T> class Matrix {
public:
	Matrix() : nrow_(0), ncol_(0) { }
	Matrix(SizeType nrow, SizeType ncol) : nrow_(nrow), ncol_(ncol), data_(nrow * ncol, T(0)) { }
	void resize(SizeType nrow, SizeType ncol)
	{
		nrow_ = nrow;
		ncol_ = ncol;
		data_.assign(nrow * ncol, T(0));
	}
	SizeType n_row() const { return nrow_; }
	SizeType n_col() const { return ncol_; }
	SizeType rows() const { return nrow_; }
	SizeType cols() const { return ncol_; }
	T& operator()(SizeType i, SizeType j) { return data_[i + j * nrow_]; } // column-major like PsimagLite::Matrix
	const T& operator()(SizeType i, SizeType j) const { return data_[i + j * nrow_]; }

private:
	SizeType nrow_, ncol_;
	std::vector<T> data_;
};

namespace detail {
// real symmetric a (n x n, row-major, destroyed): eigenvalues ascending in w, eigenvectors in the COLUMNS of a (a[i*n+k] = component i of vector k)
inline void symmetricEigen(std::vector<double>& a, SizeType n, std::vector<double>& w)
{
	std::vector<double> e(n, 0.0);
	w.assign(n, 0.0);
	if (n == 0) return;
	auto A = [&](SizeType i, SizeType j) -> double& { return a[i * n + j]; };
	// Householder reduction to tridiagonal form, transformations accumulated in a
	for (SizeType i = n - 1; i >= 1; i--) {
		const SizeType l = i - 1;
		double h = 0.0, scale = 0.0;
		if (l > 0) {
			for (SizeType k = 0; k <= l; k++) scale += std::fabs(A(i, k));
			if (scale == 0.0) {
				e[i] = A(i, l);
			} else {
				for (SizeType k = 0; k <= l; k++) {
					A(i, k) /= scale;
					h += A(i, k) * A(i, k);
				}
				double f = A(i, l);
				double g = (f >= 0.0 ? -std::sqrt(h) : std::sqrt(h));
				e[i] = scale * g;
				h -= f * g;
				A(i, l) = f - g;
				f = 0.0;
				for (SizeType j = 0; j <= l; j++) {
					A(j, i) = A(i, j) / h;
					g = 0.0;
					for (SizeType k = 0; k <= j; k++) g += A(j, k) * A(i, k);
					for (SizeType k = j + 1; k <= l; k++) g += A(k, j) * A(i, k);
					e[j] = g / h;
					f += e[j] * A(i, j);
				}
				const double hh = f / (h + h);
				for (SizeType j = 0; j <= l; j++) {
					f = A(i, j);
					e[j] = g = e[j] - hh * f;
					for (SizeType k = 0; k <= j; k++) A(j, k) -= (f * e[k] + g * A(i, k));
				}
			}
		} else {
			e[i] = A(i, l);
		}
		w[i] = h;
	}
	w[0] = 0.0;
	e[0] = 0.0;
	for (SizeType i = 0; i < n; i++) {
		if (w[i] != 0.0) {
			for (SizeType j = 0; j < i; j++) {
				double g = 0.0;
				for (SizeType k = 0; k < i; k++) g += A(i, k) * A(k, j);
				for (SizeType k = 0; k < i; k++) A(k, j) -= g * A(k, i);
			}
		}
		w[i] = A(i, i);
		A(i, i) = 1.0;
		for (SizeType j = 0; j < i; j++) A(j, i) = A(i, j) = 0.0;
	}
	// implicit QL on (w, e)
	for (SizeType i = 1; i < n; i++) e[i - 1] = e[i];
	e[n - 1] = 0.0;
	for (SizeType l = 0; l < n; l++) {
		int iter = 0;
		SizeType m;
		do {
			for (m = l; m + 1 < n; m++) {
				const double dd = std::fabs(w[m]) + std::fabs(w[m + 1]);
				if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
			}
			if (m != l) {
				if (iter++ == 300) throw RuntimeError("diag: QL iteration did not converge\n");
				double g = (w[l + 1] - w[l]) / (2.0 * e[l]);
				double r = std::hypot(g, 1.0);
				g = w[m] - w[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
				double s = 1.0, c = 1.0, p = 0.0;
				SizeType i = m;
				bool underflow = false;
				while (i-- > l) {
					double f = s * e[i], b = c * e[i];
					e[i + 1] = (r = std::hypot(f, g));
					if (r == 0.0) {
						w[i + 1] -= p;
						e[m] = 0.0;
						underflow = true;
						break;
					}
					s = f / r;
					c = g / r;
					g = w[i + 1] - p;
					r = (w[i] - g) * s + 2.0 * c * b;
					w[i + 1] = g + (p = s * r);
					g = c * r - b;
					for (SizeType k = 0; k < n; k++) {
						f = A(k, i + 1);
						A(k, i + 1) = s * A(k, i) + c * f;
						A(k, i) = c * A(k, i) - s * f;
					}
				}
				if (underflow) continue;
				w[l] -= p;
				e[l] = g;
				e[m] = 0.0;
			}
		} while (m != l);
	}
	// ascending order, columns carried along
	std::vector<SizeType> perm(n);
	for (SizeType i = 0; i < n; i++) perm[i] = i;
	std::stable_sort(perm.begin(), perm.end(), [&](SizeType x, SizeType y) { return w[x] < w[y]; });
	std::vector<double> w2(n), a2(n * n);
	for (SizeType k = 0; k < n; k++) {
		w2[k] = w[perm[k]];
		for (SizeType i = 0; i < n; i++) a2[i * n + k] = a[i * n + perm[k]];
	}
	w.swap(w2);
	a.swap(a2);
}
} // namespace detail

// diag(m, eigs, 'V'): on return the columns of m are the orthonormal eigenvectors, eigs ascending (the PsimagLite call of DefaultSymmetry.h:86)
inline void diag(Matrix<double>& m, std::vector<double>& eigs, char)
{
	const SizeType n = m.n_row();
	if (m.n_col() != n) throw RuntimeError("diag: matrix not square\n");
	std::vector<double> a(n * n);
	for (SizeType i = 0; i < n; i++)
		for (SizeType j = 0; j < n; j++) a[i * n + j] = 0.5 * (m(i, j) + m(j, i));
	detail::symmetricEigen(a, n, eigs);
	for (SizeType i = 0; i < n; i++)
		for (SizeType k = 0; k < n; k++) m(i, k) = a[i * n + k];
}

inline void diag(Matrix<std::complex<double>>& m, std::vector<double>& eigs, char)
{
	const SizeType n = m.n_row(), N = 2 * n;
	if (m.n_col() != n) throw RuntimeError("diag: matrix not square\n");
	std::vector<double> a(N * N);
	for (SizeType i = 0; i < n; i++)
		for (SizeType j = 0; j < n; j++) {
			const std::complex<double> h = 0.5 * (m(i, j) + std::conj(m(j, i)));
			a[i * N + j] = a[(i + n) * N + (j + n)] = h.real();
			a[(i + n) * N + j] = h.imag();
			a[i * N + (j + n)] = -h.imag();
		}
	std::vector<double> w;
	detail::symmetricEigen(a, N, w);
	// every eigenvalue comes twice ((u;v) and (-v;u) both stand for z = u + iv): keep n complex vectors that are
	// orthonormal under the complex inner product
	eigs.assign(n, 0.0);
	std::vector<std::vector<std::complex<double>>> kept;
	std::vector<double> keptw;
	for (SizeType k = 0; k < N && kept.size() < n; k++) {
		std::vector<std::complex<double>> z(n);
		for (SizeType i = 0; i < n; i++) z[i] = std::complex<double>(a[i * N + k], a[(i + n) * N + k]);
		for (SizeType q = 0; q < kept.size(); q++) {
			if (std::fabs(keptw[q] - w[k]) > 1e-9 * (1.0 + std::fabs(w[k]))) continue;
			std::complex<double> ov(0.0, 0.0);
			for (SizeType i = 0; i < n; i++) ov += std::conj(kept[q][i]) * z[i];
			for (SizeType i = 0; i < n; i++) z[i] -= ov * kept[q][i];
		}
		double nrm = 0.0;
		for (SizeType i = 0; i < n; i++) nrm += std::norm(z[i]);
		if (nrm < 0.25) continue; // the partner of a vector already kept
		nrm = 1.0 / std::sqrt(nrm);
		for (SizeType i = 0; i < n; i++) z[i] *= nrm;
		kept.push_back(z);
		keptw.push_back(w[k]);
	}
	if (kept.size() != n) throw RuntimeError("diag: could not separate the Hermitian eigenvectors\n");
	for (SizeType k = 0; k < n; k++) {
		eigs[k] = keptw[k];
		for (SizeType i = 0; i < n; i++) m(i, k) = kept[k][i];
	}
}

template <typename T> Matrix<T> toDense(const CrsMatrix<T>& s)
{
	Matrix<T> m(s.rows(), s.cols());
	for (SizeType i = 0; i < s.rows(); i++)
		for (int64_t k = s.getRowPtr(i); k < s.getRowPtr(i + 1); k++) m(i, (SizeType)s.getCol(k)) += s.getValue(k);
	return m;
}

// SparseRow: add(col,value) collects; finalize sorts by column (stable), sums duplicates in
// insertion order, keeps explicit zeros ([PsimagLite] behaviour restated; call sites HubbardHelper.h:88-99).
template <typename T> class SparseRow {
public:
	void add(SizeType col, const T& v) { e_.emplace_back(col, v); }
	void clear() { e_.clear(); }
	// appends the merged row to (cols, vals); returns the number of entries written
	SizeType finalize(std::vector<int32_t>& cols, std::vector<T>& vals)
	{
		std::stable_sort(e_.begin(), e_.end(), [](const std::pair<SizeType, T>& a, const std::pair<SizeType, T>& b) { return a.first < b.first; });
		SizeType n = 0;
		for (SizeType k = 0; k < e_.size(); k++) {
			if (n > 0 && (SizeType)cols.back() == e_[k].first) {
				vals.back() += e_[k].second;
			} else {
				cols.push_back((int32_t)e_[k].first);
				vals.push_back(e_[k].second);
				n++;
			}
		}
		e_.clear();
		return n;
	}

private:
	std::vector<std::pair<SizeType, T>> e_;
};

} // namespace LppHost
#endif
