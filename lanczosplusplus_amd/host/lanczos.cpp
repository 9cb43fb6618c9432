// lanczos.cpp -- driver with the reference's command line for this path (src/lanczos.cpp:99-226,
// src/Engine/LanczosDriver1.h:47-66): reads an InputNg-style file (the reference's TestSuite inputs work
// unmodified for the in-scope models), builds the model, runs the GPU engine, prints "Energy=".
//   lanczos -f input.inp [-p precision] [-d device]
// SolverOptions=useComplex selects complex<double> (lanczos.cpp:194-226).  Observables (-g, -c, -m, ...)
// are out of scope.
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>

#include "../../include/lpp_comm_rccl.h"
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

static int envInt(const char* name, int dflt)
{
	const char* s = getenv(name);
	return s ? atoi(s) : dflt;
}

static void rcclCheck(lpp_status st)
{
	if (st != LPP_OK) throw std::runtime_error(std::string("lpp_comm_rccl: ") + lpp_rccl_last_error() + "\n");
}

// Rank 0 creates the id and publishes it atomically (write + rename); the others poll for the file.  The file is
//   "LPPRCCL1" | 8-byte launch nonce | 128-byte id
// and a reader accepts it only with ITS OWN launch nonce (a hash of LPP_RCCL_NONCE, else of the launcher's
// TORCHELASTIC_RUN_ID / MASTER_ADDR:MASTER_PORT, which every rank of one launch shares) and only when it is not older than
// the reader itself by more than five minutes: a file left behind by a crashed run is never joined.  Rank 0 unlinks any old
// file before it writes and removes the new one once ncclCommInitRank has returned (a collective: every rank has read it).
static const char kIdMagic[8] = { 'L', 'P', 'P', 'R', 'C', 'C', 'L', '1' };

static uint64_t launchNonce()
{
	std::string key;
	if (const char* s = getenv("LPP_RCCL_NONCE")) key = s;
	else {
		for (const char* name : { "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "SLURM_JOB_ID", "SLURM_STEP_ID" })
			if (const char* v = getenv(name)) key += std::string(name) + "=" + v + ";";
	}
	if (key.empty()) return 0; // no launch identity: shareUniqueId refuses more than one rank
	uint64_t h = 1469598103934665603ull; // FNV-1a
	for (unsigned char c : key) h = (h ^ c) * 1099511628211ull;
	return h ? h : 1;
}

static void shareUniqueId(char* id, int rank, int world)
{
	const char* path = getenv("LPP_RCCL_ID_FILE");
	if (world > 1 && !path) throw std::runtime_error("lanczos -P: set LPP_RCCL_ID_FILE to a path every rank can read\n");
	const uint64_t nonce = launchNonce();
	// without a launch identity every launch would share one nonce and a rank could join the id file a crashed run left behind
	if (world > 1 && nonce == 0)
		throw std::runtime_error("lanczos -P: no launch identity (set LPP_RCCL_NONCE to a value unique to this launch, or start the ranks through a launcher that exports MASTER_ADDR/MASTER_PORT, TORCHELASTIC_RUN_ID or SLURM_JOB_ID)\n");
	if (rank == 0) {
		rcclCheck(lpp_rccl_unique_id(id));
		if (world == 1) return;
		(void)unlink(path); // never leave a previous run's id where a reader could find it
		const std::string tmp = std::string(path) + ".tmp";
		FILE* f = fopen(tmp.c_str(), "wb");
		const bool ok = f && fwrite(kIdMagic, 1, 8, f) == 8 && fwrite(&nonce, 1, 8, f) == 8 && fwrite(id, 1, LPP_RCCL_ID_BYTES, f) == LPP_RCCL_ID_BYTES;
		if (f) fclose(f);
		if (!ok) throw std::runtime_error("lanczos -P: cannot write the RCCL id file\n");
		if (rename(tmp.c_str(), path) != 0) throw std::runtime_error("lanczos -P: cannot publish the RCCL id file\n");
		return;
	}
	const time_t started = time(nullptr);
	for (int tries = 0; tries < 6000; tries++) { // up to 10 minutes
		FILE* f = fopen(path, "rb");
		if (f) {
			char magic[8];
			uint64_t got = 0;
			struct stat sb;
			const bool fresh = fstat(fileno(f), &sb) == 0 && sb.st_mtime + 300 >= started;
			const bool ok = fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kIdMagic, 8) == 0 && fread(&got, 1, 8, f) == 8 && got == nonce
			    && fread(id, 1, LPP_RCCL_ID_BYTES, f) == LPP_RCCL_ID_BYTES;
			fclose(f);
			if (ok && fresh) return;
		}
		usleep(100000);
	}
	throw std::runtime_error("lanczos -P: timed out waiting for this launch's RCCL id file (stale files are ignored)\n");
}

static long binomial(long n, long k)
{
	if (k < 0 || k > n) return 0;
	long r = 1;
	for (long i = 1; i <= k; i++) r = r * (n - k + i) / i;
	return r;
}

// one process per GPU: this rank's rows are assembled on its GPU, the Lanczos loop runs on all ranks in lock step
// (every rank takes bitwise-identical decisions), rank 0 prints the reference's "Energy=" line
static double realPart(double v) { return v; }
static double imagPart(double) { return 0.0; }
static double realPart(const std::complex<double>& v) { return v.real(); }
static double imagPart(const std::complex<double>& v) { return v.imag(); }

// Models without a device assembler for row partitions (Heisenberg, TjMultiOrb, ...): every rank lets the model assemble its CSR on the host
// (DefaultSymmetry.h:54-57), keeps rows [rank * per, (rank + 1) * per) and hands them over with lpp_engine_set_csr_partition; the Lanczos
// vector travels by all-gather (the north_star's form).
template <typename ComplexOrRealType>
static int partitionedFromHostCsr(const ModelBase<ComplexOrRealType>& model, const ParametersForSolver<double>& params, int precision, bool isComplex)
{
	const int rank = envInt("RANK", 0), world = envInt("WORLD_SIZE", 1), local = envInt("LOCAL_RANK", rank);
	typename ModelBase<ComplexOrRealType>::SparseMatrixType h;
	model.setupHamiltonian(h);
	const int64_t rows = (int64_t)h.rows(), per = (rows + world - 1) / world;
	std::vector<int64_t> starts((size_t)world + 1);
	for (int k = 0; k <= world; k++) starts[(size_t)k] = std::min<int64_t>((int64_t)k * per, rows);
	const int64_t r0 = starts[(size_t)rank], r1 = starts[(size_t)rank + 1], p0 = h.rowptr()[(size_t)r0];
	std::vector<int64_t> rp((size_t)(r1 - r0) + 1);
	for (int64_t r = r0; r <= r1; r++) rp[(size_t)(r - r0)] = h.rowptr()[(size_t)r] - p0;
	char id[LPP_RCCL_ID_BYTES];
	shareUniqueId(id, rank, world);
	lpp_config cfg;
	lpp_config_default(&cfg);
	cfg.device = local;
	cfg.dtype = isComplex ? LPP_C128 : LPP_F64;
	cfg.max_steps = (int32_t)params.steps;
	cfg.min_steps = (int32_t)params.minSteps;
	cfg.eps = params.tolerance;
	cfg.reortho = params.options.find("reortho") != LppHost::String::npos;
	cfg.save_vectors = 0;
	EngineHandle engine(cfg);
	lpp_rccl_comm* comm = nullptr;
	rcclCheck(lpp_rccl_comm_create(&comm, rank, world, id, local, lpp_engine_stream(engine.get()), per, (int32_t)params.steps, isComplex ? 1 : 0, 0));
	if (rank == 0 && world > 1) (void)unlink(getenv("LPP_RCCL_ID_FILE"));
	rcclCheck(lpp_rccl_comm_selftest(comm));
	lppCheck(lpp_engine_set_csr_partition(engine.get(), lpp_rccl_comm_get(comm), rows, starts.data(), rp.data(), h.colind().data() + p0, h.values().data() + p0));
	double e0 = 0;
	lpp_stats st;
	lppCheck(lpp_engine_lanczos(engine.get(), nullptr, 1, &e0, nullptr, &st));
	if (rank == 0) {
		std::cout.precision(precision);
		model.print(std::cout);
		std::cout << "Energy=" << e0 << "\n";
		std::cerr << "#LanczosSteps=" << st.steps << " rows=" << rows << " ranks=" << world << " exchange=" << (world > 1 ? "allgather" : "none") << " (host CSR, row partition)\n";
	}
	lppCheck(lpp_engine_sync(engine.get()));
	rcclCheck(lpp_rccl_comm_destroy(comm));
	return 0;
}

template <typename ComplexOrRealType> static int mainPartitioned(LppHost::InputReadable& io, int precision, bool onthefly)
{
	typedef LppHost::Geometry<ComplexOrRealType> GeometryType;
	const bool isComplex = sizeof(ComplexOrRealType) == 2 * sizeof(double); // SolverOptions=useComplex (lanczos.cpp:194-226): complex vectors, hoppings with phases
	const int rank = envInt("RANK", 0), world = envInt("WORLD_SIZE", 1), local = envInt("LOCAL_RANK", rank);
	GeometryType geometry(io);
	ModelSelector<ComplexOrRealType> modelSelector(io, geometry);
	const ModelBase<ComplexOrRealType>& model = modelSelector();
	const HubbardOneOrbital<ComplexOrRealType>* hub = dynamic_cast<const HubbardOneOrbital<ComplexOrRealType>*>(&model);
	if (!hub) {
		if (onthefly) throw std::runtime_error("lanczos -P: SolverOptions=InternalProductOnTheFly is the Hubbard family's path\n");
		ParametersForSolver<double> paramsGeneric(io, "Lanczos");
		return partitionedFromHostCsr<ComplexOrRealType>(model, paramsGeneric, precision, isComplex);
	}
	const int n = (int)geometry.numberOfSites();
	const typename ModelBase<ComplexOrRealType>::BasisBaseType::PairIntType parts = model.basis().parts();
	const long n_up = binomial(n, parts.first), n_dn = binomial(n, parts.second);
	const long per = (n_dn + world - 1) / world;
	std::string exchange = hub->jCoupling() ? "allgather" : "transpose"; // spin-flip terms (SuperHubbardExtended) need the gathered vector
	if (const char* s = getenv("LPP_EXCHANGE")) exchange = s;
	const long chunk = exchange == "transpose" ? (long)lpp_xchg_chunk(n_up, n_dn, world) : 0; // up range per rank rounded to 16: product-basis kernels
	ParametersForSolver<double> params(io, "Lanczos");
	char id[LPP_RCCL_ID_BYTES];
	// everything that can throw on ONE rank comes before the communicator exists: afterwards a rank that leaves alone would
	// strand its peers inside a collective
	if (onthefly && hub->jCoupling()) throw std::runtime_error("lanczos -P: the term-list product of Model=SuperHubbardExtended runs on one GPU: use the stored engine (SolverOptions=none) with -P\n");
	shareUniqueId(id, rank, world);
	lpp_config cfg;
	lpp_config_default(&cfg);
	cfg.device = local;
	cfg.dtype = isComplex ? LPP_C128 : LPP_F64;
	cfg.max_steps = (int32_t)params.steps;
	cfg.min_steps = (int32_t)params.minSteps;
	cfg.eps = params.tolerance;
	cfg.reortho = params.options.find("reortho") != LppHost::String::npos;
	cfg.save_vectors = 0; // energies only: no vector of the full length is ever gathered
	EngineHandle engine(cfg); // engine-owned stream; the communicator is told which one below
	lpp_rccl_comm* comm = nullptr;
	rcclCheck(lpp_rccl_comm_create(&comm, rank, world, id, local, lpp_engine_stream(engine.get()), per * n_up, (int32_t)params.steps, isComplex ? 1 : 0, chunk));
	if (rank == 0 && world > 1) (void)unlink(getenv("LPP_RCCL_ID_FILE")); // ncclCommInitRank is a collective: every rank has read the id
	rcclCheck(lpp_rccl_comm_selftest(comm)); // every callback once, checked on every rank, before any step depends on them
	std::vector<double> hr((size_t)n * n), hi((size_t)n * n);
	for (int k = 0; k < n * n; k++) {
		hr[(size_t)k] = realPart(hub->hoppings()[(size_t)k]);
		hi[(size_t)k] = imagPart(hub->hoppings()[(size_t)k]);
	}
	const double* him = isComplex ? hi.data() : nullptr;
	if (onthefly)
		lppCheck(lpp_engine_setup_hubbard_onthefly_ext(engine.get(), lpp_rccl_comm_get(comm), n, parts.first, parts.second, hr.data(), him,
		                                               hub->hubbardU.data(), hub->potentialEffective.data(), hub->coulombCoupling()));
	else
		lppCheck(lpp_engine_assemble_hubbard_super(engine.get(), lpp_rccl_comm_get(comm), n, parts.first, parts.second, hr.data(), him,
		                                           hub->hubbardU.data(), hub->potentialEffective.data(), hub->coulombCoupling(), hub->jCoupling()));
	double e0 = 0;
	lpp_stats st;
	lppCheck(lpp_engine_lanczos(engine.get(), nullptr, 1, &e0, nullptr, &st));
	if (rank == 0) {
		std::cout.precision(precision);
		model.print(std::cout);
		std::cout << "Energy=" << e0 << "\n";
		std::cerr << "#LanczosSteps=" << st.steps << " rows=" << model.size() << " ranks=" << world << " exchange=" << (world > 1 ? exchange : "none") << "\n";
	}
	lppCheck(lpp_engine_sync(engine.get())); // the communicator's buffers are released next: nothing of the engine may be queued on them
	rcclCheck(lpp_rccl_comm_destroy(comm));
	return 0;
}

int main(int argc, char** argv)
{
	LppHost::String file;
	int device = 0, precision = 8, opt = 0;
	bool partitioned = false;
	while ((opt = getopt(argc, argv, "f:p:d:P")) != -1) {
		switch (opt) {
		case 'f': file = optarg; break;
		case 'p': precision = atoi(optarg); break;
		case 'd': device = atoi(optarg); break;
		case 'P': partitioned = true; break;
		default: std::cerr << "USAGE: " << argv[0] << " -f filename [-p precision] [-d device] [-P]\n"; return 1;
		}
	}
	if (file.empty()) {
		std::cerr << "USAGE: " << argv[0] << " -f filename [-p precision] [-d device] [-P]\n";
		return 1;
	}
	try {
		LppHost::InputReadable io(file);
		LppHost::String options("none");
		if (io.has("SolverOptions=")) io.readline(options, "SolverOptions=");
		const bool onthefly = options.find("InternalProductOnTheFly") != LppHost::String::npos;
		const bool isComplex = options.find("useComplex") != LppHost::String::npos;
		if (partitioned) return isComplex ? mainPartitioned<std::complex<double>>(io, precision, onthefly) : mainPartitioned<double>(io, precision, onthefly);
		return isComplex ? mainLoop0<std::complex<double>>(io, device, precision, onthefly) : mainLoop0<double>(io, device, precision, onthefly);
	} catch (std::exception& e) {
		std::cerr << "lanczos: " << e.what();
		return 2;
	}
}
