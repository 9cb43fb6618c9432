// test_boundary.cpp -- exercises the parts of the drop-in boundary the `lanczos` driver does not reach (tests/test_host_shim.py):
//   sectors : Engine::computeAllStatesBelow's loop over symmetry sectors (Engine.h:616-652) with a toy two-sector symmetry
//             class in the shape of ReflectionSymmetry (one CSR per sector, setPointer selects, transform embeds);
//   fulldiag: the catch -> hamiltonian.fullDiag fallback (Engine.h:627-639, DefaultSymmetry.h:80-93);
//   decomp  : the reference's two-argument InternalProductStored constructor + LanczosSolver(params).decomposition
//             (Engine.h:472-478) with LanczosSteps= taken from the input.
//   test_boundary -f input.inp -m sectors|fulldiag|decomp [-p precision]
#include <getopt.h>

#include <cstdlib>
#include <iostream>

#include "EngineGpu.h"

using namespace LanczosPlusPlus;

// Block-diagonal toy symmetry: sector 0 = H + 3, sector 1 = H, sector 2 = empty.  The ground state lives in sector 1;
// an engine that keeps solving sector 0's matrix reports E0 + 3.
template <typename BasisType_, typename GeometryType_> class ToySectorSymmetry {
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

	ToySectorSymmetry(const BasisType&, const GeometryType&, LppHost::String) : pointer_(0) { }
	template <typename SomeModelType> void init(const SomeModelType& model, const BasisType& basis)
	{
		SparseMatrixType h;
		model.setupHamiltonian(h, basis);
		sectors_.resize(3);
		sectors_[1] = h;
		sectors_[0] = h;
		for (SizeType i = 0; i < h.rows(); i++)
			for (int64_t k = h.getRowPtr(i); k < h.getRowPtr(i + 1); k++)
				if ((SizeType)h.getCol(k) == i) sectors_[0].values()[k] += ComplexOrRealType(3.0);
		sectors_[2].resize(0, 0);
		total_ = 2 * h.rows();
	}
	void fullDiag(VectorRealType& eigs, MatrixType& fm) const
	{
		fm = LppHost::toDense(sectors_[pointer_]);
		LppHost::diag(fm, eigs, 'V');
	}
	// ReflectionSymmetry::transform: the sector vector becomes a vector of the whole space
	void transform(VectorVectorType& vs, SizeType offset)
	{
		for (VectorType& v : vs) {
			VectorType full(total_, ComplexOrRealType(0));
			std::copy(v.begin(), v.end(), full.begin() + offset);
			v.swap(full);
		}
	}
	SizeType sectors() const { return sectors_.size(); }
	void setPointer(SizeType p) { pointer_ = p; }
	LppHost::String name() const { return "toy"; }
	SizeType rows() const { return sectors_[pointer_].rows(); }
	const SparseMatrixType& storedMatrix() const { return sectors_[pointer_]; }
	void releaseHostMatrix() { }

private:
	std::vector<SparseMatrixType> sectors_;
	SizeType pointer_, total_ = 0;
};

template <typename ComplexOrRealType> int run(LppHost::InputReadable& io, const LppHost::String& mode, int precision)
{
	typedef LppHost::Geometry<ComplexOrRealType> GeometryType;
	typedef ModelBase<ComplexOrRealType> ModelType;
	typedef typename ModelType::BasisBaseType BasisType;
	GeometryType geometry(io);
	ModelSelector<ComplexOrRealType> modelSelector(io, geometry);
	const ModelType& model = modelSelector();
	std::cout.precision(precision);
	if (mode == "sectors") {
		typedef ToySectorSymmetry<BasisType, GeometryType> SymmetryType;
		Engine<ModelType, InternalProductStored, SymmetryType> engine(model, io, 0);
		std::cout << "Energy=" << engine.energies(0) << "\n";
		// where the state sits in the whole (two-block) space: everything outside [offset, offset+n) must be zero
		const auto& z = engine.eigenvector(0);
		double inside = 0, outside = 0;
		const SizeType n = model.size();
		for (SizeType i = 0; i < z.size(); i++) (i >= n && i < 2 * n ? inside : outside) += LppHost::real(z[i] * LppHost::conj(z[i]));
		std::cout << "Sector=" << engine.sector() << " Length=" << z.size() << " NormInside=" << inside << " NormOutside=" << outside << "\n";
		return 0;
	}
	if (mode == "fulldiag") {
		typedef DefaultSymmetry<BasisType, GeometryType> SymmetryType;
		Engine<ModelType, InternalProductStored, SymmetryType> engine(model, io, 0);
		std::cout << "Energy=" << engine.energies(0) << "\n";
		std::cout << "UsedFullDiag=" << (engine.usedFullDiag() ? 1 : 0) << "\n";
		SizeType excited = 0;
		if (io.has("Excited=")) io.readline(excited, "Excited=");
		for (SizeType k = 0; k <= excited; k++) std::cout << "Level" << k << "=" << engine.energies(k) << "\n";
		return 0;
	}
	if (mode == "decomp") {
		// exactly the reference's sequence (Engine.h:186-187,472-478): symmetry, two-argument InternalProduct, params, solver
		typedef DefaultSymmetry<BasisType, GeometryType> SymmetryType;
		typedef InternalProductStored<ModelType, SymmetryType> InternalProductType;
		typedef ParametersForSolver<double> ParametersForSolverType;
		typedef std::vector<ComplexOrRealType> VectorType;
		typedef LanczosSolver<ParametersForSolverType, InternalProductType, VectorType> LanczosSolverType;
		SymmetryType rs(model.basis(), model.geometry(), "");
		InternalProductType hamiltonian(model, rs);
		ParametersForSolverType params(io, "Lanczos");
		LanczosSolverType lanczosSolver(hamiltonian, params);
		VectorType init(hamiltonian.rows());
		fillRandom(init);
		typename LanczosSolverType::TridiagonalMatrixType ab;
		lanczosSolver.decomposition(init, ab);
		std::cout << "Steps=" << ab.size() << "\n";
		for (SizeType j = 0; j < ab.size(); j++) std::cout << "ab " << j << " " << ab.a(j) << " " << ab.b(j) << "\n";
		return 0;
	}
	std::cerr << "unknown mode " << mode << "\n";
	return 1;
}

int main(int argc, char** argv)
{
	LppHost::String file, mode("sectors");
	int precision = 12, opt = 0;
	while ((opt = getopt(argc, argv, "f:m:p:")) != -1) {
		switch (opt) {
		case 'f': file = optarg; break;
		case 'm': mode = optarg; break;
		case 'p': precision = atoi(optarg); break;
		default: return 1;
		}
	}
	if (file.empty()) return 1;
	try {
		LppHost::InputReadable io(file);
		LppHost::String options("none");
		if (io.has("SolverOptions=")) io.readline(options, "SolverOptions=");
		const bool isComplex = options.find("useComplex") != LppHost::String::npos;
		return isComplex ? run<std::complex<double>>(io, mode, precision) : run<double>(io, mode, precision);
	} catch (std::exception& e) {
		std::cerr << "test_boundary: " << e.what();
		return 2;
	}
}
