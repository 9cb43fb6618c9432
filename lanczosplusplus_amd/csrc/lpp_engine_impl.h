// lpp_engine_impl.h -- internal state of an lpp_engine (shared by the .hip translation units).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "lpp_host.h"
#include "lpp_kernels.h"
#include "lpp_pb_kernels.h"
#include "lpp_pbig_kernels.h"
#include "lpp_tj.h"

#define HIP_TRY(expr)                                                                                                  \
	do {                                                                                                               \
		hipError_t _err = (expr);                                                                                      \
		if (_err != hipSuccess)                                                                                        \
			return lpp::fail(LPP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_err));                          \
	} while (0)

#define HIP_TRY_MEM(expr)                                                                                              \
	do {                                                                                                               \
		hipError_t _err = (expr);                                                                                      \
		if (_err != hipSuccess)                                                                                        \
			return lpp::fail(_err == hipErrorOutOfMemory ? LPP_ERR_NOMEM : LPP_ERR_HIP,                                  \
			                 std::string(#expr) + ": " + hipGetErrorString(_err));                                       \
	} while (0)

namespace lpp {

// device-resident CSR (+ optional sliced layout used by k_spmv_sliced)
struct DevCsr {
	int64_t nrows = 0, nnz = 0;
	int64_t* rowptr = nullptr;
	int32_t* col = nullptr;
	void* val = nullptr;
	bool owned = true;
	int G = 16; // lanes per row of the row-group kernel
	bool sliced = false;
	bool window = false; // LDS-window kernel (K3)
	int tmpl = 0; // block-periodic structure (k_tmpl_check): 1 = slice_ptr / row_len / scol describe block 0 only, 2 = codes / code_ptr too
	uint32_t* tw = nullptr; // packed padded copy of a level-2 template (k_tmpl_pack), with tw_off / tw_len per template slice
	int32_t *tw_off = nullptr, *tw_len = nullptr;
	uint8_t* dcode = nullptr; // diagonal split off the per-row entries: dictionary code(s) per row
	int64_t code_words = 0; // 32-bit words in `codes`
	bool local16 = false; // scol holds 16-bit window-local columns (window kernel, every per-row entry inside its block)
	bool outs_first = false; // plain format, window kernel: a row's entries that leave its row block are stored in its first slots (k_slice_fill)
	int pad = 0; // ... and the vectors are pitched: row block b starts at element b * (geom.B + pad), the stored columns are pitched positions
	int64_t hint_block = 0; // natural row block of the basis (N_up), 0 = unknown
	int64_t src_elems = 0; // length of the vector the columns index (0 = nrows)
	SliceGeom geom {};
	int64_t* slice_ptr = nullptr;
	int32_t* row_len = nullptr;
	int32_t* scol = nullptr;
	void* sval = nullptr;
	// value dictionary (coded layout): sval == nullptr, values = dict[codes]
	bool coded = false;
	uint32_t* codes = nullptr;
	int64_t* code_ptr = nullptr;
	double* dict = nullptr;
	int ndict = 0;
	// shared-offset entries (k_dia_split): the sliced arrays then hold the "rest" CSR described by rrowptr
	bool no_dia = false;
	bool known_sorted = false; // rows strictly sorted by column by construction (device assembler): skip the check
	int64_t* rrowptr = nullptr; // row pointers of the rest CSR (null: nothing was split off)
	int64_t rnnz = 0; // its entries
	int64_t ndia = 0; // shared entries, summed over slices
	int dia_stride = 0; // places per slice in dia_off / dia_val
	int32_t* dia_off = nullptr;
	void* dia_val = nullptr;
	// split-panel layout (k_split_count): this CSR holds the in-block entries; the entries that leave the row block live in
	// *out_part, whose rows are the panel-major reordering given by out_rowmap (row r' of the out part is row out_rowmap[r'])
	// (several of them, by source block range: split_parts, see SplitParams)
	DevCsr* out_part = nullptr; // array of split_parts matrices
	int split_parts = 0;
	int32_t* out_rowmap = nullptr;
	int64_t split_B = 0, split_nb = 0, split_pblk = 0;
	int64_t out_nnz() const
	{
		int64_t n = 0;
		for (int q = 0; q < split_parts; q++) n += out_part[q].nnz;
		return n;
	}
};

// matrix-free Hubbard product state: two one-species matrices instead of the full CSR
struct KronState {
	bool active = false;
	DevCsr up; // sliced H_up
	DevCsr dn; // plain H_down
	uint32_t *up_words = nullptr, *dn_words = nullptr;
	double* U = nullptr;
	double *cdiag_up = nullptr, *cdiag_dn = nullptr, *cross = nullptr; // Coulomb term of HubbardOneBandExtended (null: none)
	int L = 0;
	int64_t n_up = 0, n_dn = 0, id0 = 0, nid = 0;
	bool window = false;
	double equiv_nnz = 0; // nnz of the stored CSR this product stands for
	// packed H_up (col16 | code8 | code8), see lpp_kron_kernels.h
	bool packed = false;
	uint32_t* pk_words = nullptr;
	int32_t *pk_off = nullptr, *pk_len = nullptr;
	double* pk_dict = nullptr;
	int pk_spb = 0;
	int pk_nchunk = 1, pk_cw = 0; // LDS window pieces of the packed H_up (N_up beyond LDS: see k_spmv_kron_chunked)
	// term-list form (k_asm_apply): nothing but the term list and the couplings live on the device; every row re-derives its
	// entries per product.  Serves Model=SuperHubbardExtended, whose spin-flip terms move both species.
	bool terms = false;
	void* terms_params = nullptr; // AsmParams (host copy; its pointers are the device buffers below)
	void* terms_bufs[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
};

// product-basis stored matrix  H = 1 (x) T + C (x) 1 + D  (lpp_pb_kernels.h): everything the 5.8e9-entry CSR of BASELINE
// config 2 needs is T, C, a 256-entry dictionary and one code per row
struct PbState {
	bool active = false;
	int64_t n_up = 0, n_blk = 0, pitch = 0;
	int64_t nnz = 0; // entries of the CSR this stands for
	// in-block matrix T (off-diagonal part): packed for k_pb_up + plain CSR for lpp_engine_get_csr
	int G = 0;
	double gval[8] = { 0 };
	int spb = 0;
	uint32_t* tw = nullptr;
	int32_t* tw_off = nullptr;
	uint16_t* tw_len = nullptr;
	int64_t tw_words = 0, t_entries = 0, t_slots = 0;
	int64_t* t_ptr = nullptr;
	int32_t* t_col = nullptr;
	double* t_val = nullptr;
	// block couplings C (off-diagonal)
	int64_t* c_ptr = nullptr;
	int32_t* c_col = nullptr;
	uint8_t* c_code = nullptr;
	int64_t c_nnz = 0;
	int rowcap = 0, ids_per_wg = 0, down_grid = 0;
	int down_rounds = 1, ids_per_round = 0; // k_pb_down: pieces of a workgroup's block range per panel (one LDS image each)
	void* down_image = nullptr; // rounds > 1: the images, prepared once (k_pb_down_image)
	size_t down_lds = 0;
	int32_t* order = nullptr; // blocks of every workgroup's range by decreasing list length
	int* pace = nullptr;
	double* z = nullptr; // alpha C y of the product in flight (n_blk * pitch doubles)
	double* u = nullptr; // alpha (T y + D y) of the product in flight
	double* xy = nullptr; // device scalar: Re<x|y> left by the last combine pass = the next step's <y | x_old>
	// several GPUs, transposition exchange (pb_tx_*): this rank holds the in-block part for its own blocks [blk0, blk0 + nblk_loc)
	// (vectors, u and dcode cover those only) and applies the block couplings to the received transposed slice: all n_blk blocks,
	// rows of pitch_dn positions (= the up-index range of this rank, a multiple of 16)
	bool tx = false;
	int64_t blk0 = 0, nblk_loc = 0, pitch_dn = 0;
	int64_t nnz_loc = 0; // entries of the CSR rows this rank holds (== nnz on one GPU)
	// chained scale-free steps (pb_launch_chain): the pass r_{j+1} = w_j - g r_j of the last step has not been run yet;
	// ycur holds w_j, xcur holds r_j, g = *pend_a / *pend_b2
	bool pending = false;
	const double* pend_a = nullptr;
	const double* pend_b2 = nullptr;
	// rows beyond one LDS window / vectors beyond 4 GiB (lpp_pbig_kernels.h)
	bool big = false; // in-block part by pieces of W positions (k_pb_up_big)
	bool big2 = false; // ... two blocks per workgroup and template read (k_pb_up_big2)
	int W = 0, npieces = 1;
	uint32_t* fw = nullptr; // entries that leave their piece
	int32_t* f_off = nullptr;
	uint16_t* f_len = nullptr;
	int64_t f_words = 0, f_entries = 0;
	// positions of a block stored in the order of their list lengths (single-GPU, one-window form): stored position p holds the
	// basis state perm[p] of the species, inv[perm[p]] = p.  Null: natural order.  Only the boundary knows (vec_from_host / _to_host,
	// the start vector, the diagonal's assembly, lpp_engine_get_csr); every kernel of a step is position-blind.
	int32_t* perm = nullptr;
	int32_t* inv = nullptr;
	// rows beyond one LDS window, segmented form (lpp_pbseg.h, k_pb_up_seg): T decomposed by the high sites of the species' basis word;
	// the stored order of the positions (perm / inv) is then the segments by length
	bool seg = false;
	bool seg_one = false; // one block per workgroup (pb_chain)
	// a chain (pb_chain) is planned from the model's parameters alone: no CSR is ever held, lpp_engine_get_csr re-runs the device assembler
	// from this host copy of what lpp_engine_assemble_heisenberg was given
	bool chain_model = false;
	int chain_L = 0, chain_m = 0, chain_nfield = 0;
	std::vector<double> chain_jpm, chain_jzz, chain_field;
	int seg_nitems = 0, seg_nsegs = 0, seg_ws = 0, seg_wmax = 0, seg_nc = 0, seg_nh = 0, seg_pre0 = 4;
	int64_t seg_bytes = 0; // description of T held on the device
	void* seg_items = nullptr;
	void* seg_segs = nullptr;
	void* seg_cross = nullptr;
	void* seg_hh = nullptr;
	void* seg_slices = nullptr;
	uint32_t* seg_tw = nullptr;
	uint32_t* seg_xw = nullptr;
	// complex hoppings: the vectors are complex, a block holds 2 n_c real positions (re, im interleaved; n_up, pitch count doubles), T is
	// stored REALIFIED (row 2i: (2c, Re t), (2c+1, -Im t); row 2i+1: (2c, Im t), (2c+1, Re t)) so that k_pb_up is the real kernel, the couplings
	// keep complex values (cdict, k_pb_down<CPLX>), the diagonal code of a position is stored for both of its doubles.  t_ptr / t_col / t_val
	// keep the complex T of n_c rows ((re, im) pairs) for lpp_engine_get_csr.
	bool cplx = false;
	int64_t n_c = 0; // complex positions per block
	double* cdict = nullptr; // 256 complex coupling values
	int pre0 = 4; // chained step, two value groups: chunks of group 0 requested one slice ahead (group 1: 8 - pre0)
	bool wide = false; // vector beyond 4 GiB: k_pb_down<WIDE> (64-bit addresses from line numbers)
	bool parts = false; // couplings over parts of the source range (k_pb_down_parts; 64-bit addresses too)
	int nparts = 1, ent_cap = 0, pace_stride = 0, maxr = 0;
	int64_t part_blocks = 0;
	int32_t* c_pstart = nullptr;
	// diagonal
	double* dict = nullptr; // 256 doubles
	int ndict = 0;
	uint8_t* dcode = nullptr; // n_blk * pitch codes
	double* dval = nullptr; // the diagonal as a plain f64 stream (n_blk * pitch doubles) when it has more than 256 distinct values; the codes are then all 0
	int64_t* blockbase = nullptr; // first CSR entry of every block (n_blk + 1), for get_csr
};

// one-orbital t-J Hamiltonian without a stored matrix, hole-major order (lpp_tj_kernels.h)
struct TjState {
	bool active = false;
	TjModel model;
	int Lo = 0, lb = 0, nhi = 0, nlo = 0, ns = 0, nblk = 0, chunks = 0, grid = 0, kbits = 0;
	int64_t pitch = 0; // elements between blocks
	int64_t nnz = 0; // entries of the CSR this stands for
	int64_t table_bytes = 0;
	bool cplx_hops = false;
	uint32_t* pat = nullptr;
	int32_t* hi_base = nullptr;
	uint16_t* lo_rank = nullptr;
	void *blocks = nullptr, *pairs = nullptr, *hops = nullptr, *items = nullptr;
	int32_t* order = nullptr;
	double* diag = nullptr; // nblk * pitch
	int32_t* perm = nullptr; // nblk * ns: stored (block, pattern) -> index in the reference's basis
};

// what lpp_engine_set_model_* said about the matrix the NEXT lpp_engine_set_csr hands over (kind 0: nothing)
struct ModelHint {
	int kind = 0; // 1: one-orbital t-J (tj), 2: S = 1/2 Heisenberg (L, m, jpm, jzz, field)
	TjModel tj;
	int L = 0, m = 0, nfield = 0;
	std::vector<double> jpm, jzz, field;
};

} // namespace lpp

struct lpp_engine {
	lpp_config cfg {};
	int is_complex = 0;
	size_t esz = 8;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int spmv_max_blocks = 4096;
	int64_t row_block_hint = 0; // lpp_engine_set_row_block
	int k2_variant = 4; // bit1: XCD-contiguous map, bit2: 8 slots per batch (set in lpp_engine_create)
	int num_cus = 256;

	// matrix: A_loc has columns inside this rank's slice, A_rem (multi-GPU only) indexes the gathered buffer
	lpp::DevCsr A_loc, A_rem;
	lpp::KronState kron;
	lpp::PbState pb;
	lpp::TjState tj;
	lpp::ModelHint hint;
	// pitched vector layout (product-basis matrices): block b of `pitch_rows` valid elements starts at element b*pitch; 0 = contiguous
	int64_t pitch = 0, pitch_rows = 0, pitch_blocks = 0;
	// transposition exchange (multi-GPU Hubbard): A_loc = diagonal + up-hops on the rank's slice, A_rem = down-hops
	// on the UP-partitioned transposed slice; see lpp_assemble.hip / one_step
	bool tx = false;
	int64_t tx_per = 0, tx_peru = 0;
	int64_t kron_n_up_tx = 1; // N_up of the transposition layout
	bool has_matrix() const { return A_loc.rowptr != nullptr || kron.active || pb.active || tj.active; }
	int64_t n_local = 0, n_global = 0, row_start = 0;
	double spmv_bytes = 0;

	// communicator (copied); has_comm false on the single-GPU path
	lpp_comm comm {};
	bool has_comm = false;

	// work vectors (doubles, padded to an even count)
	int64_t nd = 0, nd_pad = 0, n2 = 0;
	double *x = nullptr, *y = nullptr;
	double* V = nullptr; // Krylov basis, column j at V + j*ldv
	int64_t ldv = 0; // in doubles
	int vcap = 0; // columns allocated
	double* zwork = nullptr; // two-pass Ritz accumulators

	// scalars (a_j, b_j^2, reortho coefficients, scratch): engine-owned or inside comm.red_buf
	int M = 0;
	double* scal_own = nullptr;
	double *ab_dev = nullptr, *coef_dev = nullptr, *tmp_dev = nullptr;
	int ab_off = 0, coef_off = 0, tmp_off = 0;
	double* partial = nullptr;
	double* h_scal = nullptr; // pinned mirror of ab_dev: 2*M doubles

	// Lanczos run state
	bool active = false;
	bool saving = false; // Lanczos vectors kept in V
	int step = 0; // steps enqueued so far
	double* ycur = nullptr; // current Lanczos vector (V column or e->y)
	double* xcur = nullptr; // accumulator vector of the recurrence (e->x, or e->y/e->x alternating when scale-free)
	bool scalefree = false; // unnormalised Lanczos vectors, scalings folded into the SpMV epilogue (no swap pass)
	std::vector<hipEvent_t> step_events;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> spmv_events;
	std::vector<int> spmv_event_cols; // per bracket: 0 = a product, > 0 = a blocked Gram-Schmidt call against that many columns
	size_t spmv_events_used = 0;
	hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
	lpp_stats stats {};

	// layout inside the scalar buffer: ab[2j] = a_j, ab[2j+1] = b_j^2 (j < M); coef: 2 doubles per Krylov
	// column (re, im) at [2M, 4M); 8 scratch doubles at [4M, 4M+8)
	void bind_scalars(double* base)
	{
		ab_off = 0;
		coef_off = 2 * M;
		tmp_off = 4 * M;
		ab_dev = base + ab_off;
		coef_dev = base + coef_off;
		tmp_dev = base + tmp_off;
	}
	lpp_status adopt_comm(const lpp_comm* c);
	void collect_spmv_times();
};

namespace lpp {
struct AsmParams; // lpp_assemble_kernels.h
void free_csr(DevCsr& A);
lpp_status finalize_csr(lpp_engine* e, DevCsr& A, bool allow_drop_plain, int force_mode = 0, int64_t force_block = 0);
void free_kron(lpp_engine* e);
void drop_product(lpp_engine* e); // matrix-free state, split flags: called by every matrix setup
int kron_launch(lpp_engine* e, const void* ywin, const void* ydown, void* x, double* partial, const EpiScale& sc = EpiScale { nullptr, nullptr, 0 }, int part = 0,
                int64_t b0 = 0, int64_t cnt = -1); // blocks [b0, b0+cnt) of the slice (parts 0 and 1)
void set_spmv_bytes(lpp_engine* e);
lpp_status alloc_work(lpp_engine* e);
int spmv_launch(lpp_engine* e, const DevCsr& A, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc = EpiScale { nullptr, nullptr, 0 });
// product-basis layout (lpp_pb.hip)
void free_pb(lpp_engine* e);
// complex hoppings (PbState::cplx): pb_build then gets the REALIFIED in-block matrix (2 n_c rows) as t_rp / t_ci / t_va and the couplings'
// structure as c_rp / c_ci (c_va unused); the complex matrices themselves come through this
struct PbCplxInput {
	int64_t n_c; // complex positions per block
	const int64_t* t_rp; // the complex in-block matrix, n_c rows, values as (re, im) pairs
	const int32_t* t_ci;
	const double* t_va;
	const double* c_va; // complex coupling values, (re, im) pairs, aligned with c_ci
};
// complex in-block matrix (n rows, (re, im) pairs, diagonal skipped) -> the realified one of 2n rows (PbState::cplx)
void pb_realify(int64_t n, const int64_t* rp, const int32_t* ci, const double* va, std::vector<int64_t>& rrp, std::vector<int32_t>& rci, std::vector<double>& rva);
// T and C as host CSRs over one species each (diagonal entries are ignored), sorted 256-entry dictionary holding every coupling
// value; the caller fills pb.dcode (n_blk*pitch codes) afterwards
lpp_status pb_build(lpp_engine* e, int64_t n_up, int64_t n_blk, const int64_t* t_rp, const int32_t* t_ci, const double* t_va,
                    const int64_t* c_rp, const int32_t* c_ci, const double* c_va, const double* dict256, int ndict,
                    int64_t blk0 = 0, int64_t nblk_loc = -1, int64_t pitch_dn = 0, int64_t nblk_padded = 0, const struct PbCplxInput* cx = nullptr,
                    struct SegPlan* pre = nullptr, int64_t pre_nnz = 0);
// A matrix whose off-diagonal part is the hopping matrix of ONE species on a chain -- the S = 1/2 Heisenberg chain in the S_z basis
// (Heisenberg.h:278-307: S+S- moves an up spin, nothing sits between neighbours) -- as ONE block of the product-basis form: the in-block
// kernel k_pb_up_seg and the streaming pass, no couplings.  Planned from the model alone (P: the assembler's parameters of the same model,
// device pointers valid during the call; hv[to * L + from] != 0: the amplitudes in the planner's convention): no CSR is built.  The layout
// is checked by one product against the assembler's row walk (k_asm_apply) before it is used.
lpp_status pb_chain(lpp_engine* e, const AsmParams& P, int L, int n, const std::vector<double>& hv, bool* done);
// the plain CSR of the S = 1/2 Heisenberg model in the reference's order, nothing else (lpp_assemble.hip; lpp_engine_get_csr of a chain)
lpp_status assemble_heisenberg_raw(lpp_engine* e, int L, int m, const double* jpm, const double* jzz, const double* field, int nfield, DevCsr& A);
int pb_launch(lpp_engine* e, const void* y, void* x, double* partial, const EpiScale& sc = EpiScale { nullptr, nullptr, 0 }, bool defer_combine = false);
// the streaming pass of the scale-free Lanczos step on a product-basis matrix: x = beta x + u + z - (a/b2_prev) y, |x|^2 partials
bool pb_chain_ok(const lpp_engine* e);
void pb_tx_up(lpp_engine* e, const void* y, const EpiScale& sc, int64_t b0, int64_t cnt);
void pb_tx_down(lpp_engine* e, const void* gath, void* send2, const EpiScale& sc);
int pb_tx_unpack_combine(lpp_engine* e, void* x, const void* y, const void* recv2, const EpiScale& sc, int64_t chunk, double* partial, const double* shift);
int pb_launch_chain(lpp_engine* e, void* w, void* y, double* partial, const EpiScale& sc, const double* g_a, const double* g_b2, const double* shift);
void pb_materialise(lpp_engine* e, void* y, const void* x, const double* g_a, const double* g_b2, double* partial);
int pb_combine_axpy(lpp_engine* e, void* x, const void* y, const EpiScale& sc, const double* a_ptr, const double* b2_prev, double* partial);
lpp_status pb_get_csr(lpp_engine* e, int64_t* rowptr, int32_t* colind, void* values);
// a handed-over CSR of product-basis form -> the product-basis layout (verified row by row); *done == false: keep the general layout
lpp_status pb_from_csr(lpp_engine* e, const DevCsr& A, int64_t n_up, bool* done);
int64_t pb_pitch_for(int64_t n_up);
// t-J without a stored matrix (lpp_tj.hip).  tj_build: P = the assembler's parameters of the same model (device pointers valid during the
// call); *done == false: the layout does not apply, the caller assembles the CSR
void free_tj(lpp_engine* e);
bool tj_applies(const lpp_engine* e, const TjModel& M);
lpp_status tj_build(lpp_engine* e, const TjModel& M, const AsmParams& P, bool* done);
int tj_launch(lpp_engine* e, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc);
lpp_status tj_vec_from_host(lpp_engine* e, double* dev, const void* host);
lpp_status tj_vec_to_host(lpp_engine* e, void* host, const double* dev);
void tj_fill_random(lpp_engine* e, double* dev, uint64_t seed);
lpp_status assemble_tj_raw(lpp_engine* e, const TjModel& M, DevCsr& A); // the plain CSR in the reference's order (lpp_assemble.hip)
// A CSR handed over together with a description of its model (lpp_engine_set_model_*; lpp_assemble.hip): the device assembler regenerates the
// matrix from the description, and only if that is the handed-over CSR bit for bit is the model taken in its structured form (the t-J model
// without a stored matrix, a spin chain as one block of the segmented form).  *done == false: the caller keeps the CSR's general layout.
lpp_status model_layout_from_hint(lpp_engine* e, const DevCsr& A, bool* done);
// host <-> device vector copies that know the pitched layout
lpp_status vec_from_host(lpp_engine* e, double* dev, const void* host);
lpp_status vec_to_host(lpp_engine* e, void* host, const double* dev);
void vec_fill_random(lpp_engine* e, double* dev, uint64_t seed);
} // namespace lpp
