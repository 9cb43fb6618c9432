// lanczos.cpp -- driver with the reference's command line for this path (src/lanczos.cpp:99-226,
// src/Engine/LanczosDriver1.h:47-66): reads an InputNg-style file (the reference's TestSuite inputs work
// unmodified for the in-scope models), builds the model, runs the GPU engine, prints "Energy=".
//   lanczos -f input.inp [-p precision] [-d device]
// SolverOptions=useComplex selects complex<double> (lanczos.cpp:194-226).  Observables (-g, -c, -m, ...)
// are out of scope.
#include <getopt.h>

#include <cstdlib>
#include <iostream>

#include "EngineGpu.h"

using namespace LanczosPlusPlus;

// mainLoop3 (LanczosDriver1.h:47-66): build the engine, print the ground-state energy
template <typename ModelType, typename SymmetryType, template <typename, typename> class InternalProductTemplate>
int mainLoop3(const ModelType& model, LppHost::InputReadable& io, int device, int precision)
{
	typedef Engine<ModelType, InternalProductTemplate, SymmetryType> EngineType;
	std::cout.precision(precision);
	EngineType engine(model, io, device);
	std::cout << "Energy=" << engine.energies(0) << "\n";
	std::cerr << "#LanczosSteps=" << engine.lanczosSteps() << " rows=" << model.size() << "\n";
	return 0;
}

template <typename ComplexOrRealType> int mainLoop0(LppHost::InputReadable& io, int device, int precision, bool onthefly)
{
	typedef LppHost::Geometry<ComplexOrRealType> GeometryType;
	typedef ModelBase<ComplexOrRealType> ModelType;
	typedef DefaultSymmetry<typename ModelType::BasisBaseType, GeometryType> SymmetryType;
	GeometryType geometry(io);
	ModelSelector<ComplexOrRealType> modelSelector(io, geometry);
	const ModelType& model = modelSelector();
	model.print(std::cout);
	// stored / on-the-fly switch on the SolverOptions substring (LanczosDriver1.h:217-239)
	if (onthefly) return mainLoop3<ModelType, SymmetryType, InternalProductOnTheFly>(model, io, device, precision);
	return mainLoop3<ModelType, SymmetryType, InternalProductStored>(model, io, device, precision);
}

int main(int argc, char** argv)
{
	LppHost::String file;
	int device = 0, precision = 8, opt = 0;
	while ((opt = getopt(argc, argv, "f:p:d:")) != -1) {
		switch (opt) {
		case 'f': file = optarg; break;
		case 'p': precision = atoi(optarg); break;
		case 'd': device = atoi(optarg); break;
		default: std::cerr << "USAGE: " << argv[0] << " -f filename [-p precision] [-d device]\n"; return 1;
		}
	}
	if (file.empty()) {
		std::cerr << "USAGE: " << argv[0] << " -f filename [-p precision] [-d device]\n";
		return 1;
	}
	try {
		LppHost::InputReadable io(file);
		LppHost::String options("none");
		if (io.has("SolverOptions=")) io.readline(options, "SolverOptions=");
		const bool onthefly = options.find("InternalProductOnTheFly") != LppHost::String::npos;
		const bool isComplex = options.find("useComplex") != LppHost::String::npos;
		return isComplex ? mainLoop0<std::complex<double>>(io, device, precision, onthefly) : mainLoop0<double>(io, device, precision, onthefly);
	} catch (std::exception& e) {
		std::cerr << "lanczos: " << e.what();
		return 2;
	}
}
