/*
 * lpp_engine.h -- C ABI of the MI355X-native Lanczos inner engine (liblpp_engine.so).
 *
 * Drop-in boundary for ONE path of g1257/LanczosPlusPlus: the stored-CSR ground-state solve
 *   Engine::computeAllStatesBelow            reference src/Engine/Engine.h:601-657
 *     -> InternalProductStored::matrixVectorProduct   src/Engine/InternalProductStored.h:121-124
 *     -> DefaultSymmetry::matrixVectorProduct         src/Engine/DefaultSymmetry.h:112-116
 *     -> CrsMatrix::matrixVectorProduct / LanczosSolver::computeAllStatesBelow [PsimagLite]
 * The reference has no FFI; its boundary is C++ template duck-typing.  These
 * entry points are what a binding of that path needs; INTEGRATION.md shows the C++ shim
 * (InternalProductStored / LanczosSolver look-alikes) that forwards to them.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns an
 * lpp_status (0 = ok); lpp_last_error() gives the message of the last failure on the calling
 * thread.  One engine per host thread; calls block unless stated.  Host buffers stay owned
 * by the caller (the engine copies); device memory is owned by the engine unless a *_device /
 * comm buffer is handed in.  Values are f64 or interleaved (re,im) f64 pairs ("c128").
 * Row pointers are 64-bit (the reference's int row pointers overflow at 4x4 Hubbard).
 * There is NO CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef LPP_ENGINE_H
#define LPP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPP_ABI_VERSION 5 /* 5: LPP_SPMV_HOLE_MAJOR (lpp_layout.kernel of the t-J model), lpp_engine_set_model_tj / _heisenberg; 4: lpp_layout.segments, lpp_pb_seg_plan_stats; 2: lpp_layout.stream_bytes, lpp_pb_pack_template(bank_ways), set_solver / stream / _ext / _spin entry points; 3: lpp_layout.pieces / coupling_parts / diagonal_plain / chained_step / split_panel, lpp_stats.reortho_* */

typedef int32_t lpp_status;
enum {
	LPP_OK = 0,
	LPP_ERR_INVALID = 1, /* bad argument / shape mismatch */
	LPP_ERR_HIP = 2, /* HIP runtime error (no device, launch failure, ...) */
	LPP_ERR_NOMEM = 3, /* device or host allocation failed */
	LPP_ERR_NOCONV = 4, /* Lanczos / tridiagonal solve failed (caller may fall back, Engine.h:627) */
	LPP_ERR_STATE = 5, /* call sequence error (e.g. solve before set_csr) */
	LPP_ERR_COMM = 6 /* a communicator callback failed */
};

enum { LPP_F64 = 0, LPP_C128 = 1 };

/* SpMV kernel selection (0 = automatic) */
enum { LPP_SPMV_AUTO = 0, LPP_SPMV_ROWGROUP = 1, LPP_SPMV_SLICED = 2, LPP_SPMV_WINDOW = 3,
       LPP_SPMV_PRODUCT = 4, /* reported by lpp_engine_get_layout only: product-basis layout (device-assembled Hubbard) */
       LPP_SPMV_HOLE_MAJOR = 5 /* reported by lpp_engine_get_layout only (ABI 5): the one-orbital t-J model without a stored matrix -- states
                                  ordered (hole configuration, spin pattern of the occupied sites), every entry re-derived per product from
                                  the block's bonds and hole moves (csrc/lpp_tj_kernels.h); rows_per_block = spin patterns per hole configuration */ };

typedef struct lpp_engine lpp_engine;

/* Solver parameters; mirrors what the reference reads through
 * ParametersForSolver(io,"Lanczos") (Engine.h:609): LanczosSteps=, LanczosEps=, LanczosMinSteps=,
 * LanczosOptions=reortho, lotaMemory (SpinOrbital.cpp:213-216). */
typedef struct lpp_config {
	int32_t abi_version; /* must be LPP_ABI_VERSION */
	int32_t device; /* HIP device ordinal */
	int32_t dtype; /* LPP_F64 | LPP_C128 */
	int32_t max_steps; /* LanczosSteps (default 200) */
	int32_t min_steps; /* LanczosMinSteps (default 4) */
	int32_t reortho; /* 1: blocked CGS2 against the on-device Krylov basis every step */
	int32_t save_vectors; /* lotaMemory: 1 keep Lanczos vectors in HBM, 0 two-pass Ritz vectors, -1 auto */
	int32_t check_lag; /* steps the GPU may run ahead of the host convergence test (default 2) */
	int32_t spmv_kernel; /* LPP_SPMV_* */
	int32_t time_kernels; /* 1: bracket every SpMV launch with HIP events (for bench/roofline) */
	double eps; /* LanczosEps (default 1e-12); <= 0 disables the convergence test */
	uint64_t seed; /* seed of the built-in start vector (used when init == NULL) */
	void* stream; /* hipStream_t to run on; NULL = engine-owned stream */
	int32_t compress_values; /* lossless 8-bit value dictionary when the matrix has <= 256 distinct doubles: -1 auto, 0 off, 1 on */
	int32_t reserved;
} lpp_config;

typedef struct lpp_stats {
	int32_t steps; /* Lanczos steps that define the returned tridiagonal matrix */
	int32_t steps_enqueued; /* including run-ahead steps discarded by the lagged test */
	int32_t converged;
	int32_t vectors_saved;
	int64_t nrows, nnz;
	double seconds_total; /* wall time of the solve call */
	double spmv_ms_total; /* sum of event-timed SpMV launches (time_kernels=1) */
	int64_t spmv_launches;
	double spmv_bytes; /* algorithmic bytes of one SpMV: Z(s+4) + (N+1)8 + 3Ns */
	double reortho_ms_total; /* sum of event-timed blocked Gram-Schmidt calls (time_kernels=1, reortho=1): both CGS2 passes of a step */
	int64_t reortho_calls;
	double reortho_columns; /* Krylov columns orthogonalised against, summed over the timed calls */
} lpp_stats;

/* How a stored matrix is laid out in HBM (introspection for tests, bench.py and DESIGN.md; no effect on results). */
typedef struct lpp_layout {
	int32_t kernel; /* LPP_SPMV_ROWGROUP / _SLICED / _WINDOW actually selected */
	int32_t coded; /* 1: values stored as 8-bit dictionary codes */
	int32_t local16; /* 1: per-row columns stored as 16-bit block-local indices */
	int32_t shared_stride; /* places per 64-row slice for shared-offset entries (0: none) */
	int32_t block_template; /* block-periodic structure: 1 = row lengths and local columns stored once for all row blocks, 2 = value codes too */
	int32_t diagonal_codes; /* 1: the diagonal travels as one dictionary code per row, apart from the per-row entries */
	int64_t nnz; /* entries of the CSR this layout represents */
	int64_t per_row_entries; /* entries kept per row */
	int64_t shared_entries; /* entries stored once per slice, summed over slices */
	int64_t rows_per_block; /* row block of the sliced layout */
	int64_t resident_bytes; /* device bytes held for this matrix */
	int64_t stream_bytes; /* bytes of matrix data ONE product has to read at least once (arrays kept only for
	                         lpp_engine_get_csr and block-0 templates that stay in L2 are not streamed);
	                         stream_bytes + 3*rows*sizeof(value) is the least HBM traffic of x += H y in this layout */
	int32_t pieces; /* product-basis layout: LDS-window pieces a block's row is cut into (1: the whole row fits one window) */
	int32_t coupling_parts; /* product-basis layout: parts of the source-block range the coupling kernel walks per panel (1: whole panel) */
	int32_t diagonal_plain; /* product-basis layout: 1 = the diagonal is a plain f64 stream (more than 256 distinct values), 0 = one code per row */
	int32_t chained_step; /* product-basis layout: 1 = a scale-free Lanczos step is the chained pair of launches (k_pb_up<CHAIN> + k_pb_down<RMW>),
	                         0 = product kernels + one streaming pass that also applies the recurrence update */
	int32_t split_panel; /* general layout: > 0 = the entries that leave the row blocks are held apart in that many matrices (by source block
	                        range), rows in panel-major order (16 positions of every block, then the next 16), each applied by a launch of
	                        its own that gathers from L2 */
	int32_t rows_by_list_length; /* product-basis layout (was `reserved`, always 0, until round 3): 1 = inside a block the positions are stored in the
	                                order of their in-block list lengths, not in the basis order (slices of rows with equal lists: fewer template
	                                slots).  Internal: vectors, lpp_engine_get_csr and the start vector keep the basis order at the boundary */
	int32_t segments; /* product-basis layout, rows beyond one LDS window (ABI 4): > 0 = the in-block matrix is held decomposed by the high sites of
	                     the species' basis word into that many segments (k_pb_up_seg; positions stored segment by segment, longest first -- internal
	                     as above); 0 = one per-position template for the whole row */
	int32_t coupling_rounds; /* product-basis layout (was `reserved2`, always 0, until round 5): pieces of a workgroup's block range the coupling
	                            kernel walks inside every panel, one LDS image of coupling lists each (1: one image per launch; > 1: sectors of
	                            65536 blocks and more); 0: not a product-basis layout */
} lpp_layout;

/* Communicator for the 1-D row-partitioned multi-GPU path.  The engine never
 * opens sockets itself: the host (torch.distributed over RCCL in bench.py) supplies the
 * collectives and owns the exchange buffers.
 *   send_buf : device, shard_stride elements, 16-byte aligned (this rank's slice of the Lanczos vector, zero padded)
 *   gath_buf : device, nranks*shard_stride elems  (slice r at gath_buf + r*shard_stride)
 *   red_buf  : device doubles, red_len >= 6*(max_steps+2)   (a_j, b_j^2 and reortho coefficients)
 * allgather_begin may return before the gather completes (so the local-column SpMV overlaps
 * it); allgather_end makes the engine stream wait for it.  All callbacks return 0 on success. */
typedef struct lpp_comm {
	int32_t rank, nranks;
	void* ctx;
	void* send_buf;
	void* gath_buf;
	double* red_buf;
	int64_t shard_stride;
	int32_t red_len;
	int32_t (*allgather_begin)(void* ctx);
	int32_t (*allgather_end)(void* ctx);
	int32_t (*allreduce_sum)(void* ctx, int32_t offset, int32_t count); /* in place on red_buf[offset..] */
	/* Optional transposition exchange (lpp_engine_assemble_hubbard only; all four NULL/0 = all-gather path).
	 * Two all-to-alls of nranks equal chunks of xchg_chunk elements each:
	 *   which 0: chunk p of send_buf  -> rank p, received into chunk (sender) of gath_buf
	 *   which 1: chunk p of send2_buf -> rank p, received into chunk (sender) of recv2_buf
	 * send_buf, gath_buf, send2_buf, recv2_buf then hold nranks*xchg_chunk elements (zero-initialised by the owner).
	 * xchg_chunk = ceil(N_down/nranks) * peru, peru = up indices per rank >= ceil(N_up/nranks).  With peru rounded up to a
	 * multiple of 16 a real Hubbard matrix takes the product-basis kernels on both parts of the product;
	 * helper: lpp_xchg_chunk(). */
	void* send2_buf;
	void* recv2_buf;
	int64_t xchg_chunk;
	int32_t (*exchange_begin)(void* ctx, int32_t which);
	int32_t (*exchange_end)(void* ctx, int32_t which);
} lpp_comm;

const char* lpp_last_error(void);
int32_t lpp_abi_version(void);
void lpp_config_default(lpp_config* cfg);

lpp_status lpp_engine_create(lpp_engine** out, const lpp_config* cfg);
lpp_status lpp_engine_destroy(lpp_engine* e);
/* the hipStream_t the engine enqueues on (lpp_config.stream, or the engine-owned one): what a communicator
 * (include/lpp_comm_rccl.h) has to order its collectives against */
void* lpp_engine_stream(lpp_engine* e);

/* Solver parameters after creation: the reference builds its LanczosSolver(hamiltonian, params) AFTER the InternalProduct that
 * owns the matrix (Engine.h:608-610), so the shim's LanczosSolver pushes ParametersForSolver (LanczosSteps=, LanczosMinSteps=,
 * LanczosEps=, LanczosOptions=reortho, lotaMemory) into the engine here.  Not while a Lanczos run is active; with a
 * communicator attached comm.red_len must cover the new max_steps.  The matrix stays resident. */
lpp_status lpp_engine_set_solver(lpp_engine* e, int32_t max_steps, int32_t min_steps, double eps, int32_t reortho,
                                 int32_t save_vectors);

/* ---- the stored Hamiltonian (replaces DefaultSymmetry::matrixStored_, DefaultSymmetry.h:120) ---- */

/* Optional layout hint for the NEXT lpp_engine_set_csr / _set_csr_partition: the basis index is blocked in runs of
 * `rows_per_block` consecutive states whose in-block couplings dominate -- N_up for the Hubbard product basis
 * index = i_up + i_down * N_up (BasisHubbardLanczos.h:59-63).  The engine then stages one block of the source vector
 * in LDS per workgroup.  0 = unknown (default).  Results do not depend on it. */
lpp_status lpp_engine_set_row_block(lpp_engine* e, int64_t rows_per_block);

/* Optional description of the MODEL behind the next lpp_engine_set_csr / _set_csr_device (ABI 5; like lpp_engine_set_row_block a layout hint: results
 * never depend on it).  The reference hands its matrix over as a CSR (DefaultSymmetry.h:54-57 -> InternalProductStored.h:116); for two model
 * families the engine has a form that needs no stored matrix at all -- the one-orbital t-J model (TjMultiOrb.h:100-131; arguments as
 * lpp_engine_assemble_tj) and the S = 1/2 Heisenberg chain (Heisenberg.h:80-114; arguments as lpp_engine_assemble_heisenberg).  With a description
 * on record the engine regenerates the matrix from it with its device assembler, compares that with the handed-over CSR BIT FOR BIT, and only
 * if they are the same holds the model in its structured form (lpp_layout.kernel LPP_SPMV_HOLE_MAJOR / LPP_SPMV_PRODUCT) and drops the CSR;
 * any difference, a model the form does not apply to, or no room for the comparison keeps the general layout of the CSR as handed over.
 * The description is used for ONE matrix; nsites == 0 forgets it.  The C++ shim records it from the model object (host/EngineGpu.h). */
lpp_status lpp_engine_set_model_tj(lpp_engine* e, int32_t nsites, int32_t nup, int32_t ndown, const double* hop_re, const double* hop_im,
                                   const double* jpm, const double* jzz, const double* w, const double* potentialV, int32_t npot);
lpp_status lpp_engine_set_model_heisenberg(lpp_engine* e, int32_t nsites, int32_t szPlusConst, const double* jpm, const double* jzz,
                                           const double* field, int32_t nfield);

/* Upload a host CSR (copied).  rowptr[nrows+1], colind[nnz], values[nnz] of the engine dtype. */
lpp_status lpp_engine_set_csr(lpp_engine* e, int64_t nrows, const int64_t* rowptr, const int32_t* colind,
                              const void* values);
/* The same from DEVICE pointers of the engine's GPU (copied device-to-device, validated on the device): for hosts that
 * assemble on the GPU themselves (SURVEY 8(b) `_set_csr_device`).  Honours lpp_engine_set_row_block. */
lpp_status lpp_engine_set_csr_device(lpp_engine* e, int64_t nrows, const int64_t* d_rowptr, const int32_t* d_colind,
                                     const void* d_values);

/* Row-partitioned upload for rank `comm->rank`: rows [row_start, row_start+local_rows) of a
 * global_rows x global_rows matrix, colind holding GLOBAL column indices, rowptr relative to the
 * block (rowptr[0]==0).  shard_starts[nranks+1] gives every rank's first row.  The engine splits
 * the block into a local-column and a remote-column CSR (lpp_split_csr). */
lpp_status lpp_engine_set_csr_partition(lpp_engine* e, const lpp_comm* comm, int64_t global_rows,
                                        const int64_t* shard_starts, const int64_t* rowptr, const int32_t* colind,
                                        const void* values);

/* On-device assembly of the Hubbard Hamiltonian (GPU restatement of HubbardHelper::setupHamiltonian,
 * src/Models/HubbardOneOrbital/HubbardHelper.h:75-103, in the BasisHubbardLanczos ordering
 * rank(up) + rank(down)*N_up, BasisHubbardLanczos.h:59-63).  hop_re/hop_im: L*L row-major
 * hoppings_(i,j) (hop_im NULL for real); U[L]; V[L] (potentialV[i], i<L, both spins).
 * comm == NULL: whole matrix on this GPU.  Otherwise rows are partitioned at multiples of N_up
 * (shard_starts returned through comm-side helper lpp_partition_rows). */
lpp_status lpp_engine_assemble_hubbard(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup,
                                       int32_t ndown, const double* hop_re, const double* hop_im, const double* U,
                                       const double* V);

/* The same with the Coulomb term of Model=HubbardOneBandExtended (ModelSelector.h:76-80): the diagonal gains
 * 0.5 * sum_{i,j} ninj[i*L+j] (n_i,up + n_i,down)(n_j,up + n_j,down), ninj = the second geometry term (HubbardHelper.h:167-177,362-366).
 * ninj == NULL is lpp_engine_assemble_hubbard. */
lpp_status lpp_engine_assemble_hubbard_ext(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup, int32_t ndown,
                                           const double* hop_re, const double* hop_im, const double* U, const double* V,
                                           const double* ninj);

/* Model=SuperHubbardExtended (ModelSelector.h:76-80): Coulomb coupling ninj (geometry term 1) and spin coupling jcoup (term 2):
 * sum_ij J_ij/2 Sz_i Sz_j on the diagonal (HubbardHelper.h:158-165) and the spin-flip terms of setJTermOffDiagonal (:282-330).
 * Either may be NULL.  The spin-flip terms move both species, so the matrix takes the general layout and, on several GPUs, the
 * all-gather exchange.  Model=KaneMeleHubbard needs no entry point of its own: its hoppings are term 0 + term 1 (:63-66). */
lpp_status lpp_engine_assemble_hubbard_super(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup, int32_t ndown,
                                             const double* hop_re, const double* hop_im, const double* U, const double* V,
                                             const double* ninj, const double* jcoup);

/* Matrix-free Hubbard product (the GPU counterpart of SolverOptions=InternalProductOnTheFly:
 * InternalProductOnTheFly.h:120-123 -> HubbardHelper::matrixVectorProduct, HubbardHelper.h:105-134).
 * No CSR is stored: H = H_up (x) 1 + 1 (x) H_down + diag(U n_up n_down) in the BasisHubbardLanczos ordering;
 * only the two one-species matrices live in HBM.  Same arguments as lpp_engine_assemble_hubbard; every
 * solver entry point works unchanged afterwards, lpp_engine_get_csr does not -- except where one species' row fits the LDS
 * window (real hoppings, >= 32 MB per vector): there the matrix-free form is the product-basis layout (the two one-species
 * matrices + one diagonal code per row), the engine lpp_engine_assemble_hubbard builds, and lpp_engine_get_csr / _get_layout work. */
lpp_status lpp_engine_setup_hubbard_onthefly(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup,
                                             int32_t ndown, const double* hop_re, const double* hop_im, const double* U,
                                             const double* V);

lpp_status lpp_engine_setup_hubbard_onthefly_ext(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup, int32_t ndown,
                                                 const double* hop_re, const double* hop_im, const double* U, const double* V,
                                                 const double* ninj);

/* The same for Model=SuperHubbardExtended (ModelSelector.h:76-80): the reference's on-the-fly product applies setJTermOffDiagonal as well
 * (HubbardHelper.h:119-129, :282-330).  Those spin-flip terms move both species, so H is no longer H_up (x) 1 + 1 (x) H_down + D: with a
 * non-zero jcoup every row re-derives its entries from the term list per product (k_asm_apply), storing nothing -- the literal
 * counterpart of the reference's per-row walk, on one GPU.  jcoup == NULL (or all zero) is lpp_engine_setup_hubbard_onthefly_ext. */
lpp_status lpp_engine_setup_hubbard_onthefly_super(lpp_engine* e, const lpp_comm* comm, int32_t nsites, int32_t nup, int32_t ndown,
                                                   const double* hop_re, const double* hop_im, const double* U, const double* V,
                                                   const double* ninj, const double* jcoup);

/* On-device assembly of the S=1/2 Heisenberg Hamiltonian (Heisenberg.h:80-114,242-307) in the
 * BasisHeisenberg ordering (ascending words of fixed popcount, BasisHeisenberg.h:38-46). */
lpp_status lpp_engine_assemble_heisenberg(lpp_engine* e, int32_t nsites, int32_t szPlusConst, const double* jpm,
                                          const double* jzz, const double* field, int32_t nfield);

/* The same for any spin the reference's digit width holds (BasisHeisenberg.h:28-46: a state is L digits m_i + S of
 * bits = 1 + floor(log2(twiceS+1)) bits, one bit less for odd twiceS; ascending words of digit sum szPlusConst), with
 * the S+S- amplitudes of Heisenberg.h:278-307 and the single-ion anisotropy of Heisenberg.h:259.  Odd twiceS whose
 * digits the reference cannot hold (twiceS + 1 not a power of two) -> LPP_ERR_INVALID.  twiceS == 1 gives the matrix
 * of lpp_engine_assemble_heisenberg. */
lpp_status lpp_engine_assemble_heisenberg_spin(lpp_engine* e, int32_t nsites, int32_t twiceS, int32_t szPlusConst,
                                               const double* jpm, const double* jzz, const double* field, int32_t nfield,
                                               const double* anisotropy, int32_t naniso);

/* On-device assembly of the one-orbital t-J Hamiltonian (TjMultiOrb.h:100-131,586-783) in the
 * BasisTjMultiOrbLanczos ordering (sorted (down<<L)|up words without double occupancy). */
lpp_status lpp_engine_assemble_tj(lpp_engine* e, int32_t nsites, int32_t nup, int32_t ndown, const double* hop_re,
                                  const double* hop_im, const double* jpm, const double* jzz, const double* w,
                                  const double* potentialV, int32_t npot);

/* Copy the device CSR back (for parity tests of the assemblers).  Pass NULL pointers to query
 * sizes only.  which: 0 = whole/local-column part, 1 = remote-column part. */
lpp_status lpp_engine_get_csr(lpp_engine* e, int32_t which, int64_t* nrows, int64_t* nnz, int64_t* rowptr,
                              int32_t* colind, void* values);

/* ---- A1: x += H y  (InternalProductStored::matrixVectorProduct) on host buffers ---- */
lpp_status lpp_engine_spmv_acc(lpp_engine* e, void* x_inout, const void* y);

/* ---- A2/A3: the Lanczos solve (LanczosSolver::computeAllStatesBelow, Engine.h:626) ----
 * init: host start vector of local_rows elements or NULL (built-in splitmix64 vector, same as the oracle).
 * eigs[nstates]; ritz_vectors: host buffer nstates*local_rows elements or NULL. */
lpp_status lpp_engine_lanczos(lpp_engine* e, const void* init, int32_t nstates, double* eigs, void* ritz_vectors,
                              lpp_stats* stats);

/* LanczosSolver::decomposition (Engine.h:478): tridiagonal coefficients only.
 * a[max_steps], b[max_steps]; *nsteps receives the number of valid entries. */
lpp_status lpp_engine_decomposition(lpp_engine* e, const void* init, int32_t* nsteps, double* a, double* b,
                                    lpp_stats* stats);

/* ---- incremental interface (bench.py times exactly K steps with it) ---- */
lpp_status lpp_engine_lanczos_begin(lpp_engine* e, const void* init);
lpp_status lpp_engine_lanczos_step(lpp_engine* e, int32_t nsteps); /* enqueue only, no host sync */
lpp_status lpp_engine_sync(lpp_engine* e);
/* copy out the coefficients produced so far (after a sync): a[steps], b[steps] */
lpp_status lpp_engine_lanczos_coeffs(lpp_engine* e, int32_t* steps, double* a, double* b);
lpp_status lpp_engine_get_stats(lpp_engine* e, lpp_stats* stats);
/* which: 0 = local part (or the whole matrix on one GPU), 1 = remote part of a partitioned matrix */
lpp_status lpp_engine_get_layout(lpp_engine* e, int32_t which, lpp_layout* layout);

/* Time `iters` back-to-back SpMV launches (x += H y on resident vectors) with HIP events on the
 * engine stream; returns the average milliseconds per launch. */
lpp_status lpp_engine_bench_spmv(lpp_engine* e, int32_t warmup, int32_t iters, double* ms_per_launch);

/* ---- host-only helpers (no GPU needed; covered by the CPU test-suite) ---- */

/* 1-D contiguous row partition: starts[nranks+1]; boundaries are multiples of `block`
 * (N_up for the Hubbard product basis so that up-hops and the diagonal stay rank-local). */
/* xchg_chunk for the transposition exchange: ceil(n_down/nranks) * (ceil(n_up/nranks) rounded up to a multiple of 16) */
int64_t lpp_xchg_chunk(int64_t n_up, int64_t n_down, int32_t nranks);

lpp_status lpp_partition_rows(int64_t nrows, int32_t nranks, int64_t block, int64_t* starts);

/* Split a row block with global columns into local-column and remote-column CSRs.
 * Local columns become offsets into the rank's own slice; remote columns become indices into the
 * padded gather buffer (owner*shard_stride + offset).  Two-call protocol: with out pointers NULL
 * only nnz_loc / nnz_rem are returned. */
lpp_status lpp_split_csr(int32_t rank, int32_t nranks, const int64_t* shard_starts, int64_t shard_stride,
                         int64_t local_rows, const int64_t* rowptr, const int32_t* colind, const void* values,
                         int32_t elem_bytes, int64_t* nnz_loc, int64_t* nnz_rem, int64_t* rowptr_loc,
                         int32_t* colind_loc, void* values_loc, int64_t* rowptr_rem, int32_t* colind_rem,
                         void* values_rem);

/* Lowest `k` eigenvalues (and optionally eigenvectors, row-major z[j*n+i] = component j of vector i)
 * of the symmetric tridiagonal matrix (d[n], e[n-1]) -- the host part of the Lanczos loop. */
lpp_status lpp_tridiag_lowest(int32_t n, const double* d, const double* e, int32_t k, double* w, double* z);

/* Product-basis layout, host part (used by lpp_engine_assemble_hubbard; exposed for the CPU test-suite): packs the in-block
 * matrix `rows` x `rows` (CSR, diagonal entries skipped) into per-slice, per-value-group streams of 16-bit LDS window indices
 * (chunks of 4 slots = two 32-bit words per lane) with a slot assignment in which no LDS bank is asked for more than
 * `bank_ways` (1..4) different addresses per half-wave and slot.  pitch: multiple of 16, >= rows.  Two-call protocol:
 * off/len/words NULL returns sizes only (ngroups, slices, nwords).  group_values holds 8 doubles; off/len count chunks. */
lpp_status lpp_pb_pack_template(int64_t rows, int64_t pitch, const int64_t* rowptr, const int32_t* colind, const double* values,
                                int32_t* ngroups, double* group_values, int32_t* slices, int64_t* nwords, int32_t* off, uint16_t* len,
                                uint32_t* words, int64_t* entries, int64_t* slots, int32_t bank_ways);

/* Product-basis layout, rows beyond one LDS window, host part (ABI 4; exposed for the CPU test-suite): reads the species' basis (L sites,
 * n particles, ascending words: BasisOneSpin.h:53-61) and the hopping amplitudes off the in-block matrix `rows` x `rows` (CSR, diagonal
 * entries skipped; HubbardHelper.h:191-243 for one species), decomposes it by the high sites of the basis word (csrc/lpp_pbseg.h), expands
 * the packed description again and compares it with the matrix entry by entry, bit by bit.  wcap: longest segment (64..8128).
 * out[0] = 1 if the decomposition applies and reproduces the matrix, then out[1..13] = L, n, high sites, segments, items, item types,
 * longest item, bytes shared by class, bytes per segment, low-low entries of the item types, their lane-slots, value groups, window stride;
 * perm (rows int32, may be NULL): stored position -> basis index. */
lpp_status lpp_pb_seg_plan_stats(int64_t rows, const int64_t* rowptr, const int32_t* colind, const double* values, int32_t wcap, int64_t* out, int32_t* perm);

/* The t-J model's hole-major form, host part (ABI 5; exposed for the CPU test-suite): plans the form from (sites, sector, hoppings, J+-) -- hole
 * configurations, spin patterns and their ranking, work items, every configuration's bonds and moves (csrc/lpp_tj.h) -- and, given a CSR of the
 * model in the reference's basis order (BasisTjMultiOrbLanczos.h:29-42), expands the plan again exactly as the kernel walks it and compares it
 * with the CSR's off-diagonal entries, columns and value bits (the diagonal comes from the device assembler, not from the plan).
 * out[0] = 1 if the plan applies (and reproduces every row when a CSR is given; rowptr NULL: statistics only), out[1..9] = hole configurations,
 * spin patterns, low positions of a segment, work items, bonds, moves, bonds among the low positions, most bonds / moves of one configuration. */
lpp_status lpp_tj_plan_stats(int32_t nsites, int32_t nup, int32_t ndown, const double* hop_re, const double* hop_im, const double* jpm, int64_t nrows,
                             const int64_t* rowptr, const int32_t* colind, const void* values, int32_t is_complex, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* LPP_ENGINE_H */
