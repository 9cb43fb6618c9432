// EngineGpu.h -- the reference's template concepts for the stored-CSR path, backed by the C ABI
// (include/lpp_engine.h).  Class and member names follow the reference so a maintainer can map them 1:1:
//   DefaultSymmetry        src/Engine/DefaultSymmetry.h:41-116   (owns the CSR, init() -> model.setupHamiltonian)
//   InternalProductStored  src/Engine/InternalProductStored.h:93-132 (rows(), matrixVectorProduct: x += H y)
//   ParametersForSolver    [PsimagLite] as used at src/Engine/Engine.h:609 and src/SpinOrbital.cpp:213-216
//   LanczosSolver          [PsimagLite] computeAllStatesBelow / decomposition / computeOneState
//   Engine                 src/Engine/Engine.h:84-98,601-657 (only the ground-state part)
// The device boundary sits between Engine and the solver: the CSR is uploaded once by
// InternalProductStored, the whole Lanczos loop runs on the GPU.  Errors surface as std::runtime_error.
#ifndef LPP_HOST_ENGINE_GPU_H
#define LPP_HOST_ENGINE_GPU_H

#include <iostream>

#include "../../include/lpp_engine.h"
#include "Models.h"

namespace LanczosPlusPlus {

inline void lppCheck(lpp_status st)
{
	if (st != LPP_OK) throw std::runtime_error(std::string("lpp_engine: ") + lpp_last_error());
}

template <typename T> struct LppDtype {
	enum { value = LPP_F64 };
};
template <> struct LppDtype<std::complex<double>> {
	enum { value = LPP_C128 };
};

// RAII handle of one engine (one GPU)
class EngineHandle {
public:
	explicit EngineHandle(const lpp_config& cfg) : e_(nullptr) { lppCheck(lpp_engine_create(&e_, &cfg)); }
	~EngineHandle() { lpp_engine_destroy(e_); }
	lpp_engine* get() const { return e_; }

private:
	EngineHandle(const EngineHandle&);
	EngineHandle& operator=(const EngineHandle&);
	lpp_engine* e_;
};

template <typename RealType> struct ParametersForSolver {
	// labels Lanczos{Steps,Eps,MinSteps,Options}= with the given prefix; defaults as recalled from PsimagLite
	ParametersForSolver() : steps(200), minSteps(4), tolerance(1e-12), lotaMemory(true), options("none") { }
	ParametersForSolver(LppHost::InputReadable& io, const LppHost::String& prefix) : steps(200), minSteps(4), tolerance(1e-12), lotaMemory(true), options("none")
	{
		if (io.has(prefix + "Steps=")) io.readline(steps, prefix + "Steps=");
		if (io.has(prefix + "Eps=")) io.readline(tolerance, prefix + "Eps=");
		if (io.has(prefix + "MinSteps=")) io.readline(minSteps, prefix + "MinSteps=");
		if (io.has(prefix + "Options=")) io.readline(options, prefix + "Options=");
		int x = 0;
		if (io.has(prefix + "NoSaveLanczosVectors=")) {
			io.readline(x, prefix + "NoSaveLanczosVectors=");
			lotaMemory = (x == 0);
		}
	}
	SizeType steps, minSteps;
	RealType tolerance;
	bool lotaMemory;
	LppHost::String options;
};

// rows above which the reference refuses a dense diagonalisation (DefaultSymmetry.h:82)
enum { LPP_FULLDIAG_MAX_ROWS = 4900 };

template <typename BasisType_, typename GeometryType_> class DefaultSymmetry {
public:
	typedef GeometryType_ GeometryType;
	typedef typename GeometryType::ComplexOrRealType ComplexOrRealType;
	typedef typename LppHost::Real<ComplexOrRealType>::Type RealType;
	typedef LppHost::CrsMatrix<ComplexOrRealType> SparseMatrixType;
	typedef LppHost::Matrix<ComplexOrRealType> MatrixType;
	typedef std::vector<RealType> VectorRealType;
	typedef std::vector<ComplexOrRealType> VectorType;
	typedef std::vector<VectorType> VectorVectorType;
	typedef BasisType_ BasisType;

	DefaultSymmetry(const BasisType&, const GeometryType&, LppHost::String) { }
	template <typename SomeModelType> void init(const SomeModelType& model, const BasisType& basis) { model.setupHamiltonian(matrixStored_, basis); }
	// DefaultSymmetry.h:80-93: dense LAPACK diagonalisation of the stored matrix, refused above 4900 rows
	void fullDiag(VectorRealType& eigs, MatrixType& fm) const
	{
		if (rows_ > LPP_FULLDIAG_MAX_ROWS) throw LppHost::RuntimeError("fullDiag too big\n");
		if (matrixStored_.nonZeros() == 0 && rows_ > 0) throw LppHost::RuntimeError("fullDiag: the host matrix was released\n");
		fm = LppHost::toDense(matrixStored_);
		LppHost::diag(fm, eigs, 'V');
	}
	void transform(VectorVectorType&, SizeType) { }
	SizeType sectors() const { return 1; }
	void setPointer(SizeType) { }
	LppHost::String name() const { return "default"; }
	SizeType rows() const { return matrixStored_.rows(); }
	// the matrix of the sector selected by setPointer: what InternalProductStored makes resident on the GPU
	const SparseMatrixType& storedMatrix() const { return matrixStored_; }
	// after the upload the device copy is the resident one; small matrices stay on the host for the fullDiag fallback (Engine.h:633)
	void releaseHostMatrix()
	{
		rows_ = matrixStored_.rows();
		if (rows_ > LPP_FULLDIAG_MAX_ROWS) matrixStored_.resize(matrixStored_.rows(), matrixStored_.cols());
	}

private:
	SparseMatrixType matrixStored_;
	SizeType rows_ = 0;
};

// x += H y with H resident on the GPU
template <typename ModelType_, typename SpecialSymmetryType_> class InternalProductStored {
public:
	typedef ModelType_ ModelType;
	typedef SpecialSymmetryType_ SpecialSymmetryType;
	typedef typename ModelType::BasisBaseType BasisType;
	typedef typename SpecialSymmetryType::SparseMatrixType SparseMatrixType;
	typedef typename ModelType::RealType RealType;
	typedef typename ModelType::GeometryType GeometryType;
	typedef typename GeometryType::ComplexOrRealType ComplexOrRealType;
	typedef LppHost::Matrix<ComplexOrRealType> MatrixType;
	typedef std::vector<RealType> VectorRealType;
	typedef std::vector<ComplexOrRealType> VectorType;

	static lpp_config defaultConfig()
	{
		lpp_config cfg;
		lpp_config_default(&cfg);
		cfg.dtype = LppDtype<ComplexOrRealType>::value;
		return cfg;
	}
	// the reference's two constructors (InternalProductStored.h:104-117) with the default solver configuration
	InternalProductStored(const ModelType& model, SpecialSymmetryType& rs) : rs_(rs), engine_(defaultConfig()), basis_(model.basis()), model_(&model), rows_(0), sector_(0)
	{
		rs_.init(model, model.basis());
		upload();
	}
	InternalProductStored(const ModelType& model, const BasisType& basis, SpecialSymmetryType& rs) : rs_(rs), engine_(defaultConfig()), basis_(basis), model_(&basis == &model.basis() ? &model : nullptr), rows_(0), sector_(0)
	{
		rs_.init(model, basis);
		upload();
	}
	InternalProductStored(const ModelType& model, SpecialSymmetryType& rs, const lpp_config& cfg) : rs_(rs), engine_(cfg), basis_(model.basis()), model_(&model), rows_(0), sector_(0)
	{
		rs_.init(model, model.basis());
		upload();
	}
	InternalProductStored(const ModelType& model, const BasisType& basis, SpecialSymmetryType& rs, const lpp_config& cfg) : rs_(rs), engine_(cfg), basis_(basis), model_(&basis == &model.basis() ? &model : nullptr), rows_(0), sector_(0)
	{
		rs_.init(model, basis);
		upload();
	}
	SizeType rows() const { return rows_; }
	void matrixVectorProduct(VectorType& x, const VectorType& y) const
	{
		if (x.size() != rows_ || y.size() != rows_) throw std::runtime_error("InternalProductStored::matrixVectorProduct: size mismatch\n");
		lppCheck(lpp_engine_spmv_acc(engine_.get(), x.data(), y.data()));
	}
	// InternalProductStored.h:124: the matrix of symmetry sector p becomes THE matrix (ReflectionSymmetry.h:138-190 and
	// TranslationSymmetry.h:245-268 hold one CSR per sector); here it is made resident on the GPU in place of the previous one
	void specialSymmetrySector(SizeType p)
	{
		rs_.setPointer(p);
		if (p == sector_) return;
		sector_ = p;
		upload();
	}
	// InternalProductStored.h:126-130
	void fullDiag(VectorRealType& eigs, MatrixType& fm) const { rs_.fullDiag(eigs, fm); }
	lpp_engine* engine() const { return engine_.get(); }

private:
	void upload()
	{
		const SparseMatrixType& m = rs_.storedMatrix();
		rows_ = m.rows();
		// layout hint only: the Hubbard product basis is blocked in runs of N_up states (BasisHubbardLanczos.h:59-63);
		// it holds for the whole-space matrix of a one-sector symmetry, not for symmetry-adapted sector matrices
		const BasisHubbardLanczos* hb = dynamic_cast<const BasisHubbardLanczos*>(&basis_);
		const bool whole = rs_.sectors() == 1 && hb && (SizeType)hb->size() == rows_;
		lppCheck(lpp_engine_set_row_block(engine_.get(), whole ? (int64_t)hb->sizeUp() : 0));
		describeModel(rs_.sectors() == 1 && model_ && (SizeType)model_->size() == rows_);
		lppCheck(lpp_engine_set_csr(engine_.get(), (int64_t)m.rows(), m.rowptr().data(), m.colind().data(), m.values().data()));
		rs_.releaseHostMatrix(); // the device copy is the resident one (matrices small enough for fullDiag stay)
	}
	// The matrix is handed over as the CSR the model assembled (DefaultSymmetry.h:54-57).  For the two families the engine can hold without a
	// stored matrix -- the one-orbital t-J model, the S = 1/2 Heisenberg chain -- the shim also says WHICH model it is (couplings, sector): the
	// engine regenerates the matrix from that, compares it with the CSR bit for bit and only then takes the structured form (a layout hint:
	// results never depend on it; include/lpp_engine.h, lpp_engine_set_model_*).  Whole-space matrices only.
	static double partRe(double v) { return v; }
	static double partIm(double) { return 0.0; }
	static double partRe(const std::complex<double>& v) { return v.real(); }
	static double partIm(const std::complex<double>& v) { return v.imag(); }
	void describeModel(bool whole)
	{
		lppCheck(lpp_engine_set_model_tj(engine_.get(), 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0)); // forget an earlier one
		if (!whole) return;
		const int n = (int)model_->geometry().numberOfSites();
		const typename BasisType::PairIntType parts = basis_.parts();
		if (const TjMultiOrb<ComplexOrRealType>* tj = dynamic_cast<const TjMultiOrb<ComplexOrRealType>*>(model_)) {
			std::vector<double> hr((size_t)n * n), hi((size_t)n * n);
			bool cplx = false;
			for (size_t k = 0; k < hr.size(); k++) {
				hr[k] = partRe(tj->hoppings()[k]);
				hi[k] = partIm(tj->hoppings()[k]);
				cplx |= hi[k] != 0;
			}
			const std::vector<RealType>& pv = tj->potentialV;
			lppCheck(lpp_engine_set_model_tj(engine_.get(), n, (int32_t)parts.first, (int32_t)parts.second, hr.data(), cplx ? hi.data() : nullptr, tj->jpm().data(),
			                                 tj->jzz().data(), tj->w().data(), pv.size() >= (size_t)2 * n ? pv.data() : nullptr, pv.size() >= (size_t)2 * n ? (int32_t)pv.size() : 0));
		} else if (const Heisenberg<ComplexOrRealType>* hs = dynamic_cast<const Heisenberg<ComplexOrRealType>*>(model_)) {
			if (hs->twiceTheSpin() != 1 || !hs->anisotropy.empty() || sizeof(ComplexOrRealType) != sizeof(double)) return;
			lppCheck(lpp_engine_set_model_heisenberg(engine_.get(), n, (int32_t)parts.second, hs->jpm().data(), hs->jzz().data(),
			                                         hs->magneticField.empty() ? nullptr : hs->magneticField.data(), (int32_t)hs->magneticField.size()));
		}
	}
	SpecialSymmetryType& rs_;
	EngineHandle engine_;
	const BasisType& basis_;
	const ModelType* model_; // null when the basis is not the model's own (observables on N +- 1 sectors)
	SizeType rows_, sector_;
};

// x += H y without a stored matrix: SolverOptions=InternalProductOnTheFly (src/Engine/InternalProductOnTheFly.h:93-133).
// In scope for the HubbardOneOrbital family (the reference's threaded HubbardHelper::matrixVectorProduct, HubbardHelper.h:105-134).
template <typename ModelType_, typename SpecialSymmetryType_> class InternalProductOnTheFly {
public:
	typedef ModelType_ ModelType;
	typedef SpecialSymmetryType_ SpecialSymmetryType;
	typedef typename ModelType::BasisBaseType BasisType;
	typedef typename ModelType::RealType RealType;
	typedef typename ModelType::GeometryType GeometryType;
	typedef typename GeometryType::ComplexOrRealType ComplexOrRealType;
	typedef LppHost::Matrix<ComplexOrRealType> MatrixType;
	typedef std::vector<RealType> VectorRealType;
	typedef std::vector<ComplexOrRealType> VectorType;

	InternalProductOnTheFly(const ModelType& model, SpecialSymmetryType& rs, const lpp_config& cfg) : rs_(rs), engine_(cfg), rows_(model.size())
	{
		const HubbardOneOrbital<ComplexOrRealType>* hub = dynamic_cast<const HubbardOneOrbital<ComplexOrRealType>*>(&model);
		if (!hub) throw std::runtime_error("InternalProductOnTheFly: only Model=HubbardOneBand / HubbardOneBandExtended have an on-the-fly product\n");
		const SizeType n = model.geometry().numberOfSites();
		std::vector<double> hr(n * n), hi(n * n);
		for (SizeType k = 0; k < n * n; k++) {
			hr[k] = LppHost::real(hub->hoppings()[k]);
			hi[k] = LppHost::imag(hub->hoppings()[k]);
		}
		const typename BasisType::PairIntType parts = model.basis().parts();
		// Model=SuperHubbardExtended: the reference's on-the-fly lambda applies setJTermOffDiagonal too (HubbardHelper.h:119-129); the engine
		// then re-derives every row from the term list per product (lpp_engine_setup_hubbard_onthefly_super)
		lppCheck(lpp_engine_setup_hubbard_onthefly_super(engine_.get(), nullptr, (int32_t)n, parts.first, parts.second, hr.data(),
		                                                 sizeof(ComplexOrRealType) == 16 ? hi.data() : nullptr, hub->hubbardU.data(), hub->potentialEffective.data(),
		                                                 hub->coulombCoupling(), hub->jCoupling()));
	}
	SizeType rows() const { return rows_; }
	void matrixVectorProduct(VectorType& x, const VectorType& y) const
	{
		if (x.size() != rows_ || y.size() != rows_) throw std::runtime_error("InternalProductOnTheFly::matrixVectorProduct: size mismatch\n");
		lppCheck(lpp_engine_spmv_acc(engine_.get(), x.data(), y.data()));
	}
	void specialSymmetrySector(SizeType p) { rs_.setPointer(p); }
	// InternalProductOnTheFly.h:125-130: no stored matrix, hence no dense fallback
	template <typename VectorRealType, typename MatrixType> void fullDiag(VectorRealType&, MatrixType&) const
	{
		throw std::runtime_error("no fullDiag possible when on the fly\n");
	}
	lpp_engine* engine() const { return engine_.get(); }

private:
	SpecialSymmetryType& rs_;
	EngineHandle engine_;
	SizeType rows_;
};

struct TridiagonalMatrix {
	std::vector<double> a_, b_;
	void resize(SizeType n)
	{
		a_.assign(n, 0.0);
		b_.assign(n, 0.0);
	}
	SizeType size() const { return a_.size(); }
	double& a(SizeType i) { return a_[i]; }
	double& b(SizeType i) { return b_[i]; }
	const double& a(SizeType i) const { return a_[i]; }
	const double& b(SizeType i) const { return b_[i]; }
};

template <typename SolverParametersType, typename MatrixType, typename VectorType> class LanczosSolver {
public:
	typedef typename VectorType::value_type ComplexOrRealType;
	typedef typename LppHost::Real<ComplexOrRealType>::Type RealType;
	typedef std::vector<RealType> VectorRealType;
	typedef std::vector<VectorType> VectorVectorType;
	typedef TridiagonalMatrix TridiagonalMatrixType;

	// The reference hands (matrix, params) to the solver AFTER the matrix object exists (Engine.h:608-610): the parameters
	// are pushed into the engine here, so both reference-style InternalProduct constructors honour LanczosSteps= etc.
	LanczosSolver(const MatrixType& mat, const SolverParametersType& params) : mat_(mat), params_(params)
	{
		const bool reortho = params_.options.find("reortho") != LppHost::String::npos;
		lppCheck(lpp_engine_set_solver(mat_.engine(), (int32_t)params_.steps, (int32_t)params_.minSteps, params_.tolerance, reortho ? 1 : 0,
		                               params_.lotaMemory ? -1 : 0));
	}

	void computeAllStatesBelow(VectorRealType& eigs, VectorVectorType& zs, const VectorType& initial, SizeType nStates)
	{
		const SizeType n = mat_.rows();
		if (initial.size() != n) throw std::runtime_error("LanczosSolver: initial vector has wrong size\n");
		eigs.resize(nStates);
		std::vector<ComplexOrRealType> flat(n * nStates);
		lpp_stats st;
		lppCheck(lpp_engine_lanczos(mat_.engine(), initial.data(), (int32_t)nStates, eigs.data(), flat.data(), &st));
		zs.assign(nStates, VectorType(n));
		for (SizeType k = 0; k < nStates; k++) std::copy(flat.begin() + k * n, flat.begin() + (k + 1) * n, zs[k].begin());
		steps_ = st.steps;
	}
	void computeOneState(RealType& energy, VectorType& z, const VectorType& initial, SizeType excited)
	{
		VectorRealType eigs;
		VectorVectorType zs;
		computeAllStatesBelow(eigs, zs, initial, excited + 1);
		energy = eigs[excited];
		z = zs[excited];
	}
	void decomposition(const VectorType& initVector, TridiagonalMatrixType& ab)
	{
		ab.resize(params_.steps + 2); // the engine holds max_steps = params_.steps (set in the constructor) and writes at most that many
		int32_t n = 0;
		lppCheck(lpp_engine_decomposition(mat_.engine(), initVector.data(), &n, &ab.a(0), &ab.b(0), nullptr));
		ab.a_.resize(n);
		ab.b_.resize(n);
		steps_ = n;
	}
	SizeType steps() const { return steps_; }

private:
	const MatrixType& mat_;
	const SolverParametersType& params_;
	SizeType steps_ = 0;
};

// deterministic stand-in for PsimagLite::fillRandom (Engine.h:621): splitmix64 -> uniform(-0.5, 0.5), seed 1234
template <typename VectorType> void fillRandom(VectorType& v, uint64_t seed = 1234)
{
	typedef typename VectorType::value_type T;
	const SizeType ncomp = sizeof(T) / sizeof(double);
	double* p = reinterpret_cast<double*>(v.data());
	for (SizeType k = 0; k < v.size() * ncomp; k++) {
		uint64_t z = seed * 0x2545F4914F6CDD1DULL + k + 0x9E3779B97F4A7C15ULL;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
		z ^= z >> 31;
		p[k] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
	}
}

template <typename ModelType_, template <typename, typename> class InternalProductTemplate, typename SpecialSymmetryType_> class Engine {
public:
	typedef ModelType_ ModelType;
	typedef SpecialSymmetryType_ SpecialSymmetryType;
	typedef InternalProductTemplate<ModelType, SpecialSymmetryType> InternalProductType;
	typedef typename ModelType::RealType RealType;
	typedef typename ModelType::ComplexOrRealType ComplexOrRealType;
	typedef std::vector<ComplexOrRealType> VectorType;
	typedef std::vector<VectorType> VectorVectorType;
	typedef std::vector<RealType> VectorRealType;
	typedef ParametersForSolver<RealType> ParametersForSolverType;
	typedef LanczosSolver<ParametersForSolverType, InternalProductType, VectorType> LanczosSolverType;
	typedef typename LanczosSolverType::TridiagonalMatrixType TridiagonalMatrixType;

	Engine(const ModelType& model, LppHost::InputReadable& io, int device = 0) : model_(model), io_(io), device_(device)
	{
		SizeType excited = 0;
		if (io_.has("Excited=")) io_.readline(excited, "Excited=");
		computeAllStatesBelow(excited); // Engine.h:91-97
	}
	RealType energies(SizeType ind) const { return energies_[ind]; }
	const VectorType& eigenvector(SizeType ind) const { return vectors_[ind]; }
	SizeType lanczosSteps() const { return steps_; }
	SizeType sector() const { return sector_; } // the symmetry sector the returned states live in
	bool usedFullDiag() const { return usedFullDiag_; }

private:
	// What one symmetry sector yields: its lowest levels and their vectors (sector-local), from the device solver or -- when that
	// throws -- from the dense fallback of the reference (Engine.h:627-639).
	struct SectorStates {
		VectorRealType levels;
		VectorVectorType states;
	};

	SectorStates solveSector(InternalProductType& h, LanczosSolverType& solver, SizeType nstates)
	{
		const SizeType dim = h.rows();
		SectorStates out { VectorRealType(nstates), VectorVectorType(nstates, VectorType(dim)) };
		VectorType start(dim);
		fillRandom(start); // Engine.h:621
		try {
			solver.computeAllStatesBelow(out.levels, out.states, start, nstates); // Engine.h:626
			steps_ = solver.steps();
			return out;
		} catch (std::exception& ex) {
			std::cerr << "Engine: Lanczos Solver failed (" << ex.what() << ") trying exact diagonalization...\n";
		}
		if (nstates > dim) throw std::runtime_error("Engine: more states requested than the sector holds\n");
		typename InternalProductType::VectorRealType all(dim);
		typename InternalProductType::MatrixType vecs;
		h.fullDiag(all, vecs);
		for (SizeType k = 0; k < nstates; ++k) {
			out.levels[k] = all[k];
			for (SizeType r = 0; r < dim; ++r) out.states[k][r] = vecs(r, k);
		}
		usedFullDiag_ = true;
		return out;
	}

	// Engine::computeAllStatesBelow (Engine.h:601-657): every sector of the symmetry is solved for `excited` + 1 states; the sector whose
	// lowest level is the lowest overall supplies energies_ and vectors_, which the symmetry then maps back to the full basis.
	void computeAllStatesBelow(SizeType excited)
	{
		const SizeType nstates = excited + 1;
		ParametersForSolverType params(io_, "Lanczos");
		lpp_config cfg;
		lpp_config_default(&cfg);
		cfg.device = device_;
		cfg.dtype = LppDtype<ComplexOrRealType>::value;
		SpecialSymmetryType rs(model_.basis(), model_.geometry(), "");
		InternalProductType hamiltonian(model_, rs, cfg);
		LanczosSolverType lanczosSolver(hamiltonian, params); // pushes params into the engine
		energies_.assign(nstates, 0);
		vectors_.assign(nstates, VectorType());
		bool have = false;
		SizeType rowsBefore = 0, bestOffset = model_.size();
		for (SizeType sec = 0; sec < rs.sectors(); ++sec) {
			hamiltonian.specialSymmetrySector(sec);
			if (hamiltonian.rows() == 0) continue; // an empty sector takes no rows either (Engine.h:619)
			SectorStates found = solveSector(hamiltonian, lanczosSolver, nstates);
			const SizeType dim = found.states[0].size();
			if (!have || found.levels[0] < energies_[0]) { // Engine.h:641-649
				energies_.swap(found.levels);
				vectors_.swap(found.states);
				bestOffset = rowsBefore;
				sector_ = sec;
				have = true;
			}
			rowsBefore += dim;
		}
		rs.transform(vectors_, bestOffset); // Engine.h:654
		for (SizeType k = 0; k < nstates; k++) { // printEnergiesAndNorms, Engine.h:666-674
			RealType nrm = 0;
			for (const ComplexOrRealType& z : vectors_[k]) nrm += LppHost::real(z * LppHost::conj(z));
			std::cout << "E[" << k << "]=" << energies_[k] << " norm=" << nrm << "\n";
		}
	}
	const ModelType& model_;
	LppHost::InputReadable& io_;
	int device_;
	VectorRealType energies_;
	VectorVectorType vectors_;
	SizeType steps_ = 0, sector_ = 0;
	bool usedFullDiag_ = false;
};

} // namespace LanczosPlusPlus
#endif
