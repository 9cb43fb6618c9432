// dump_csr.cpp -- CPU-only helper for the test-suite: host assembly (Models.h) of the model described by an
// input file, written as raw arrays: header "nrows nnz is_complex\n" then rowptr(int64) colind(int32) values(f64[,f64]).
#include <cstdio>
#include <iostream>

#include "Models.h"

using namespace LanczosPlusPlus;

template <typename T> int run(LppHost::InputReadable& io, const char* out)
{
	LppHost::Geometry<T> geometry(io);
	ModelSelector<T> sel(io, geometry);
	LppHost::CrsMatrix<T> m;
	sel().setupHamiltonian(m);
	FILE* f = fopen(out, "wb");
	if (!f) return 3;
	fprintf(f, "%zu %zu %d\n", m.rows(), m.nonZeros(), (int)(sizeof(T) == 16));
	fwrite(m.rowptr().data(), sizeof(int64_t), m.rows() + 1, f);
	fwrite(m.colind().data(), sizeof(int32_t), m.nonZeros(), f);
	fwrite(m.values().data(), sizeof(T), m.nonZeros(), f);
	fclose(f);
	return 0;
}

int main(int argc, char** argv)
{
	if (argc < 3) {
		std::cerr << "USAGE: " << argv[0] << " input.inp out.bin\n";
		return 1;
	}
	try {
		LppHost::InputReadable io(argv[1]);
		LppHost::String options("none");
		if (io.has("SolverOptions=")) io.readline(options, "SolverOptions=");
		return options.find("useComplex") != LppHost::String::npos ? run<std::complex<double>>(io, argv[2]) : run<double>(io, argv[2]);
	} catch (std::exception& e) {
		std::cerr << "dump_csr: " << e.what();
		return 2;
	}
}
