// lpp_assemble.hip -- host side of the on-device Hamiltonian assembly (see lpp_assemble_kernels.h).
// Builds the delta-sorted term list of each model from the reference's element formulas and
// launches count -> scan -> fill.  Citations are relative to /root/reference/src.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lpp_assemble_kernels.h"
#include "lpp_engine_impl.h"
#include "lpp_kron_kernels.h"

using namespace lpp;

namespace {

struct HostProc {
	Proc p;
	int64_t delta;
};

inline uint64_t bit(int i) { return 1ull << i; }
inline uint64_t below(int i) { return bit(i) - 1; }
// bits [i, j)
inline uint64_t range_mask(int i, int j) { return below(j) & ~below(i); }

int64_t delta_of(const Proc& p)
{
	// bra - ket as integers: bits switched on minus bits switched off
	const uint64_t on = p.xmask & p.need_clear, off = p.xmask & p.need_set;
	return (int64_t)on - (int64_t)off;
}

void push(std::vector<HostProc>& v, uint64_t need_set, uint64_t need_clear, uint64_t xmask, uint64_t smask_ket, uint64_t smask_bra,
          int sign_const, double re, double im, bool real_only = false)
{
	HostProc h {};
	h.p.need_set = need_set;
	h.p.need_clear = need_clear;
	h.p.xmask = xmask;
	h.p.smask_ket = smask_ket;
	h.p.smask_bra = smask_bra;
	h.p.sign_const = sign_const;
	h.p.amp_re = re;
	h.p.amp_im = im;
	h.p.real_only = real_only ? 1 : 0;
	h.delta = delta_of(h.p);
	v.push_back(h);
}

std::vector<uint64_t> comb_table()
{
	// BasisOneSpin::doCombinatorial (BasisOneSpin.h:178-191); saturating beyond 64 bits
	std::vector<uint64_t> c((size_t)kCombDim * kCombDim, 0);
	for (int n = 0; n < kCombDim; n++) {
		c[(size_t)n * kCombDim] = 1;
		for (int m = 1; m <= n; m++) {
			const unsigned __int128 v = (unsigned __int128)c[(size_t)(n - 1) * kCombDim + m - 1] + c[(size_t)(n - 1) * kCombDim + m];
			c[(size_t)n * kCombDim + m] = v > (unsigned __int128)UINT64_MAX ? UINT64_MAX : (uint64_t)v;
		}
	}
	return c;
}

struct DevBuf {
	void* p = nullptr;
	~DevBuf()
	{
		if (p) (void)hipFree(p);
	}
};

template <int MODEL, typename T>
lpp_status run_assembly(lpp_engine* e, AsmParams P, DevCsr& A, int force_mode = 0, int64_t force_block = 0, bool raw = false)
{
	hipStream_t st = e->stream;
	const auto t_asm0 = std::chrono::steady_clock::now();
	const int64_t keep_src = A.src_elems;
	free_csr(A);
	A.src_elems = keep_src;
	A.hint_block = (MODEL == ASM_HUBBARD && P.part != 2) ? P.n_up : 0; // Hubbard product basis: one down configuration per block
	A.nrows = P.nloc;
	A.owned = true;
	A.known_sorted = true; // rows come out of the delta-sorted term list in column order
	HIP_TRY_MEM(hipMalloc(&A.rowptr, sizeof(int64_t) * (size_t)(P.nloc + 1)));
	HIP_TRY(hipMemsetAsync(A.rowptr, 0, sizeof(int64_t) * (size_t)(P.nloc + 1), st));
	const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((P.nloc + kBlock - 1) / kBlock, 1 << 20));
	if (P.nloc > 0) k_asm_count<MODEL><<<nb, kBlock, 0, st>>>(P, A.rowptr);
	// exclusive scan of nloc+1 lengths (last is 0) -> rowptr
	const int64_t n = P.nloc + 1;
	const int64_t nblk = (n + kScanChunk - 1) / kScanChunk;
	DevBuf sums, total;
	HIP_TRY_MEM(hipMalloc(&sums.p, sizeof(int64_t) * (size_t)nblk));
	HIP_TRY_MEM(hipMalloc(&total.p, sizeof(int64_t)));
	k_scan_block_sums<<<(int)nblk, kBlock, 0, st>>>(A.rowptr, n, (int64_t*)sums.p);
	k_scan_sums<<<1, kBlock, 0, st>>>((int64_t*)sums.p, nblk, (int64_t*)total.p);
	k_scan_apply<<<(int)nblk, kBlock, 0, st>>>(A.rowptr, n, (const int64_t*)sums.p, A.rowptr);
	HIP_TRY(hipGetLastError());
	int64_t nnz = 0;
	HIP_TRY(hipMemcpyAsync(&nnz, total.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	A.nnz = nnz;
	HIP_TRY_MEM(hipMalloc(&A.col, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
	HIP_TRY_MEM(hipMalloc(&A.val, sizeof(T) * (size_t)std::max<int64_t>(nnz, 1)));
	if (P.nloc > 0) k_asm_fill<MODEL, T><<<nb, kBlock, 0, st>>>(P, A.rowptr, A.col, (T*)A.val);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	if (getenv("LPP_VERBOSE"))
		fprintf(stderr, "lpp: %-28s %8.1f ms\n", "assembly (count, scan, fill)", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_asm0).count());
	if (raw) return LPP_OK; // the caller decides the layout (lpp_engine_assemble_heisenberg: pb_chain, else finalize_csr)
	return finalize_csr(e, A, true, force_mode, force_block);
}

template <int MODEL> lpp_status dispatch(lpp_engine* e, const AsmParams& P, DevCsr& A, int force_mode = 0, int64_t force_block = 0, bool raw = false)
{
	return e->is_complex ? run_assembly<MODEL, cplx>(e, P, A, force_mode, force_block, raw) : run_assembly<MODEL, double>(e, P, A, force_mode, force_block, raw);
}

lpp_status upload(hipStream_t st, DevBuf& b, const void* src, size_t bytes)
{
	HIP_TRY_MEM(hipMalloc(&b.p, std::max<size_t>(bytes, 8)));
	if (bytes) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
	return LPP_OK;
}

lpp_status finish_procs(std::vector<HostProc>& hp, std::vector<Proc>& out, int* nneg)
{
	std::stable_sort(hp.begin(), hp.end(), [](const HostProc& a, const HostProc& b) { return a.delta < b.delta; });
	for (size_t i = 0; i + 1 < hp.size(); i++)
		if (hp[i].delta == hp[i + 1].delta) return fail(LPP_ERR_INVALID, "assembly: two Hamiltonian terms map to the same bra (unsupported duplicate)");
	*nneg = 0;
	out.clear();
	for (auto& h : hp) {
		if (h.delta == 0) return fail(LPP_ERR_INVALID, "assembly: off-diagonal term with zero displacement");
		if (h.delta < 0) (*nneg)++;
		out.push_back(h.p);
	}
	return LPP_OK;
}

uint64_t binom(const std::vector<uint64_t>& c, int n, int m)
{
	if (n < 0 || m < 0 || m > n || n >= kCombDim) return 0;
	return c[(size_t)n * kCombDim + m];
}

// terms of the Hubbard hopping: c^dagger_j c_i for every ordered pair with hoppings_(i,j) != 0, both species,
// value h * doSign(ket,i) * doSign(ket^bit(i), j)   (HubbardHelper.h:205-243, ProgramGlobals.h:109-114)
void hubbard_terms(int L, const double* hop_re, const double* hop_im, std::vector<HostProc>& hp, int species_mask = 3)
{
	for (int i = 0; i < L; i++) {
		for (int j = 0; j < L; j++) {
			if (i == j) continue;
			const double hr = hop_re[i * L + j], hi = hop_im ? hop_im[i * L + j] : 0.0;
			if (hr == 0 && hi == 0) continue;
			for (int spin = 0; spin < 2; spin++) {
				if (!(species_mask & (1 << spin))) continue;
				const int sh = spin * L;
				push(hp, bit(i + sh), bit(j + sh), bit(i + sh) | bit(j + sh), (below(i) ^ below(j)) << sh, 0, i < j ? 1 : 0, hr, hi);
			}
		}
	}
}

// Spin-flip terms of SuperHubbardExtended (HubbardHelper.h:282-330): S+_i S-_j moves an up electron j -> i and a down electron i -> j,
// bra = (up ^ (bit i | bit j), down ^ (bit i | bit j)).  setJTermOffDiagonal adds it twice to a row -- from site i's loop with
// J(i,j)/4 and from site j's loop with J(j,i)/4 -- and SparseRow::finalize sums the two; the sign (jTermSign, :332-343) is the
// parity of the up AND the down electrons in [min, max) (BasisOneSpin::doSign, BasisOneSpin.h:100-119), evaluated on the ket.
void super_terms(int L, const double* jcoup, std::vector<HostProc>& hp)
{
	for (int i = 0; i < L; i++)
		for (int j = 0; j < L; j++) {
			if (i == j) continue;
			double a = 0.0, b = 0.0; // contribution of site i's loop (partner j) and of site j's loop (partner i)
			if (jcoup[i * L + j] != 0) {
				a = jcoup[i * L + j] * 0.5;
				a *= 0.5;
			}
			if (jcoup[j * L + i] != 0) {
				b = jcoup[j * L + i] * 0.5;
				b *= 0.5;
			}
			if (jcoup[i * L + j] == 0 && jcoup[j * L + i] == 0) continue;
			const int lo = std::min(i, j), hi = std::max(i, j);
			const uint64_t both = bit(i) | bit(j), rng = range_mask(lo, hi);
			const double amp = a + b; // one rounding at most, the same whichever of the two SparseRow::finalize meets first
			push(hp, bit(j) | (bit(i) << L), bit(i) | (bit(j) << L), both | (both << L), rng | (rng << L), 0, 0, amp, 0.0, true);
		}
}

// columns_are_local: every stored column index is rank-local (transposition exchange: own slice + transposed slice), so only
// the per-rank sizes -- checked by the caller -- have to fit 32 bits, not the global dimension
lpp_status common_setup(lpp_engine* e, int64_t nrows, int is_complex_input, bool columns_are_local = false)
{
	if (!e) return fail(LPP_ERR_INVALID, "assemble: null engine");
	if (is_complex_input && !e->is_complex) return fail(LPP_ERR_INVALID, "assemble: complex couplings need a c128 engine");
	if (nrows <= 0) return fail(LPP_ERR_INVALID, "assemble: empty Hilbert space");
	if (nrows > (int64_t)INT32_MAX && !columns_are_local)
		return fail(LPP_ERR_INVALID, "assemble: Hilbert space exceeds the 32-bit column range of the stored CSR (partition it with the transposition exchange, or use the matrix-free engine)");
	HIP_TRY(hipSetDevice(e->cfg.device));
	return LPP_OK;
}

// plain host copy of a small device CSR
lpp_status fetch_csr(const DevCsr& A, std::vector<int64_t>& rp, std::vector<int32_t>& ci, std::vector<double>& va, int doubles_per_value = 1)
{
	rp.resize((size_t)A.nrows + 1);
	ci.resize((size_t)std::max<int64_t>(A.nnz, 1));
	va.resize((size_t)std::max<int64_t>(A.nnz, 1) * (size_t)doubles_per_value);
	HIP_TRY(hipMemcpy(rp.data(), A.rowptr, sizeof(int64_t) * (size_t)(A.nrows + 1), hipMemcpyDeviceToHost));
	if (A.nnz) {
		HIP_TRY(hipMemcpy(ci.data(), A.col, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(va.data(), A.val, sizeof(double) * (size_t)A.nnz * (size_t)doubles_per_value, hipMemcpyDeviceToHost));
	}
	return LPP_OK;
}

// Hubbard straight into the product-basis layout (lpp_pb_kernels.h): the two one-species matrices (a few hundred KB, built
// by the same assembler with the other species empty), the distinct diagonal values and one diagonal code per row.  The
// N-row CSR (71 GB at BASELINE config 2) never exists.  *done stays false when the matrix does not qualify (the caller
// then takes the general path); P carries the full-matrix parameters.
// Several GPUs (transposition exchange): blk0 / nblk_loc = the rank's own down configurations, pitch_dn = its up-index range (the row
// length of the transposed slice, a multiple of 16), nblk_padded = down configurations in the transposed slice, padding included.
lpp_status assemble_hubbard_pb(lpp_engine* e, AsmParams P, int nup, int ndown, int64_t n_up, int64_t n_dn, const double* zeroU_dev, bool* done,
                               int64_t blk0 = 0, int64_t nblk_loc = -1, int64_t pitch_dn = 0, int64_t nblk_padded = 0, bool stored = false)
{
	*done = false;
	if (n_up < 512) return LPP_OK;
	const bool cplx = e->is_complex != 0;
	// complex hoppings (Peierls phases, KaneMele): one-window single-GPU form with the realified in-block matrix (PbState::cplx)
	if (cplx && (nblk_loc >= 0 || pitch_dn > 0 || (getenv("LPP_PB_COMPLEX") && atoi(getenv("LPP_PB_COMPLEX")) == 0))) return LPP_OK;
	if (P.d3) return LPP_OK; // spin-flip terms move both species: not of the form 1 (x) T + C (x) 1
	// LPP_PRODUCT_LAYOUT = 0: never, 1: whenever it applies; unset: from 32 MB per vector on.  Below that everything sits in
	// L2 / Infinity Cache anyway and the step is launch-bound: measured on Hubbard chains (scripts/experiments/pb_threshold.sh),
	// 8.5e5 rows 17.2k (product) vs 18.8k (general) iterations/s, 1.2e7 rows 4164 vs 3833, 1.3e8 rows 385 vs 246
	bool forced = false;
	if (const char* s = getenv("LPP_PRODUCT_LAYOUT")) {
		if (atoi(s) == 0) return LPP_OK;
		forced = true;
	}
	if (!forced && (size_t)n_up * (size_t)n_dn * sizeof(double) < ((size_t)32 << 20)) return LPP_OK;
	// complex hoppings run the any-number-of-groups in-block kernel (4 groups): measured 1659 against 1687 iterations/s (general layout) at
	// 1.2e7 states (0.19 GB per vector), 327 against 170 at 6.4e7 states (1.0 GB per vector; 0.14 against 8.6 GB resident)
	if (!forced && cplx && (size_t)n_up * (size_t)n_dn * 2 * sizeof(double) < ((size_t)512 << 20)) return LPP_OK;
	if (e->cfg.spmv_kernel != LPP_SPMV_AUTO || getenv("LPP_SPMV_KERNEL")) return LPP_OK;
	int want = e->cfg.compress_values;
	if (const char* s = getenv("LPP_COMPRESS_VALUES")) want = atoi(s);
	if (want == 0) return LPP_OK;
	for (const char* k : { "LPP_SHARED_OFFSETS", "LPP_LOCAL16", "LPP_DIAG_CODES", "LPP_BLOCK_TEMPLATE", "LPP_WINDOW_ROWS" })
		if (getenv(k)) return LPP_OK; // switches of the general layout: measure that one
	const int64_t pitch = pb_pitch_for(cplx ? 2 * n_up : n_up); // in doubles
	// rows beyond one LDS window and vectors beyond 4 GiB take the pieces / parts kernels (lpp_pbig_kernels.h); what the layout
	// cannot hold at all (more than 65535 blocks, coupling lists beyond LDS) comes back from pb_build as "does not apply"
	if (n_dn >= ((int64_t)1 << 24) || n_up >= ((int64_t)1 << 24)) return LPP_OK;
	{
		// two work vectors + the two parts of a product + one code per row must fit beside everything else
		const int64_t nloc = (nblk_loc >= 0 ? nblk_loc : n_dn) * pitch;
		size_t free_b = 0, total_b = 0;
		// slack: an eighth of what the layout needs, at least 64 MB (a flat 2 GiB made small forced cases fall to the other layout on a
		// nearly full GPU).  The general layout needs far more memory than this one, so there is nothing to fall back to: say so.
		const size_t need = (size_t)nloc * 33, slack = std::max<size_t>(need / 8, (size_t)64 << 20);
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + slack > free_b) {
			const std::string msg = "product-basis layout: " + std::to_string((need + slack) >> 20) + " MB needed, " + std::to_string(free_b >> 20) + " MB free on the device";
			if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: %s\n", msg.c_str());
			// the matrix-free engine's other kernels hold two vectors only (the (8,8) sector of the 4x5 lattice: 254 GB); a stored
			// matrix has nothing smaller to fall back to -- the general layout needs far more -- so it fails here, with the numbers
			return stored ? fail(LPP_ERR_NOMEM, msg) : LPP_OK;
		}
	}
	hipStream_t st = e->stream;
	// one-species matrices: hops of that species + its potential diagonal (ignored below; the true diagonal is per row)
	DevCsr Tm, Cm;
	struct Guard {
		DevCsr &a, &b;
		~Guard()
		{
			free_csr(a);
			free_csr(b);
		}
	} guard { Tm, Cm };
	AsmParams P1 = P;
	P1.d0 = zeroU_dev;
	P1.d2 = nullptr;
	P1.ndown = 0;
	P1.part = 0;
	P1.row0 = 0;
	P1.nup = nup;
	P1.n_up = n_up;
	P1.nrows_global = P1.nloc = n_up;
	lpp_status rc = dispatch<ASM_HUBBARD>(e, P1, Tm, LPP_SPMV_ROWGROUP, 0);
	if (rc != LPP_OK) return rc;
	P1.nup = ndown;
	P1.n_up = n_dn;
	P1.nrows_global = P1.nloc = n_dn;
	rc = dispatch<ASM_HUBBARD>(e, P1, Cm, LPP_SPMV_ROWGROUP, 0);
	if (rc != LPP_OK) return rc;
	std::vector<int64_t> trp, crp;
	std::vector<int32_t> tci, cci;
	std::vector<double> tva, cva;
	if ((rc = fetch_csr(Tm, trp, tci, tva, cplx ? 2 : 1)) != LPP_OK) return rc;
	if ((rc = fetch_csr(Cm, crp, cci, cva, cplx ? 2 : 1)) != LPP_OK) return rc;
	// distinct diagonal values over all rows
	DevBuf table, overflow;
	HIP_TRY_MEM(hipMalloc(&table.p, sizeof(unsigned long long) * kDictTable));
	HIP_TRY_MEM(hipMalloc(&overflow.p, sizeof(int)));
	HIP_TRY(hipMemsetAsync(table.p, 0xff, sizeof(unsigned long long) * kDictTable, st));
	HIP_TRY(hipMemsetAsync(overflow.p, 0, sizeof(int), st));
	AsmParams Pf = P;
	Pf.row0 = 0;
	Pf.nloc = P.nrows_global;
	Pf.part = 0;
	const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((Pf.nloc + kBlock - 1) / kBlock, 1 << 16));
	k_pb_diag_collect<ASM_HUBBARD><<<nb, kBlock, 0, st>>>(Pf, (unsigned long long*)table.p, (int*)overflow.p);
	std::vector<unsigned long long> host(kDictTable);
	int ov = 0;
	HIP_TRY(hipMemcpyAsync(host.data(), table.p, sizeof(unsigned long long) * kDictTable, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&ov, overflow.p, sizeof(int), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	// More than 256 distinct diagonal values (site-dependent hubbardU / potentialV, HubbardHelper.h:138-189: disorder) do not
	// end the layout: the diagonal then travels as ONE plain f64 stream (8 instead of 1 byte per row), added by the streaming
	// pass behind the two product kernels (PbCombineArgs::d); T and C stay what they are.  LPP_PB_PLAIN_DIAG=1 forces it (tests).
	std::vector<unsigned long long> keys;
	keys.push_back(0ull); // code 0 = +0.0 (padding places of k_pb_down)
	auto add_key = [&](unsigned long long k) {
		if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
	};
	size_t ndiag = 0;
	for (unsigned long long k : host)
		if (k != kDictEmpty) ndiag++;
	bool plain_diag = ov != 0 || ndiag > 250 || (getenv("LPP_PB_PLAIN_DIAG") && atoi(getenv("LPP_PB_PLAIN_DIAG")) != 0);
	if (plain_diag && getenv("LPP_PB_PLAIN_DIAG") && atoi(getenv("LPP_PB_PLAIN_DIAG")) == 0) return LPP_OK; // switched off: general layout
	if (!plain_diag)
		for (unsigned long long k : host)
			if (k != kDictEmpty) add_key(k);
	if (!cplx) // (complex couplings have a dictionary of their own: pb_build)
		for (int64_t b = 0; b < n_dn; b++)
			for (int64_t p = crp[(size_t)b]; p < crp[(size_t)b + 1] && keys.size() <= 256; p++)
				if (cci[(size_t)p] != b) {
					unsigned long long k;
					std::memcpy(&k, &cva[(size_t)p], 8);
					add_key(k);
				}
	if (keys.size() > 256) return LPP_OK;
	std::sort(keys.begin(), keys.end());
	std::vector<double> dict(256);
	for (size_t i = 0; i < 256; i++) std::memcpy(&dict[i], &keys[std::min(i, keys.size() - 1)], 8);
	if (cplx) {
		std::vector<int64_t> rrp;
		std::vector<int32_t> rci;
		std::vector<double> rva;
		pb_realify(n_up, trp.data(), tci.data(), tva.data(), rrp, rci, rva);
		PbCplxInput cx;
		cx.n_c = n_up;
		cx.t_rp = trp.data();
		cx.t_ci = tci.data();
		cx.t_va = tva.data();
		cx.c_va = cva.data();
		rc = pb_build(e, 2 * n_up, n_dn, rrp.data(), rci.data(), rva.data(), crp.data(), cci.data(), nullptr, dict.data(), (int)keys.size(), 0, -1, 0, 0, &cx);
	} else
	rc = pb_build(e, n_up, n_dn, trp.data(), tci.data(), tva.data(), crp.data(), cci.data(), cva.data(), dict.data(), (int)keys.size(), blk0, nblk_loc, pitch_dn,
	              nblk_padded);
	if (rc == LPP_ERR_INVALID) { // not representable (e.g. more than 8 distinct in-block values): general path
		if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: the product-basis layout does not apply: %s\n", lpp_last_error());
		free_pb(e);
		return LPP_OK;
	}
	if (rc != LPP_OK) return rc;
	if (nblk_loc >= 0) { // codes for the rank's own rows only
		Pf.row0 = blk0 * n_up;
		Pf.nloc = nblk_loc * n_up;
	}
	if (plain_diag) {
		const size_t loc = (size_t)std::max<int64_t>(e->pb.nblk_loc, 1) * (size_t)e->pb.pitch;
		HIP_TRY_MEM(hipMalloc(&e->pb.dval, sizeof(double) * loc));
		HIP_TRY(hipMemsetAsync(e->pb.dval, 0, sizeof(double) * loc, st));
		if (Pf.nloc > 0) k_pb_diag_values<ASM_HUBBARD><<<nb, kBlock, 0, st>>>(Pf, e->pb.pitch, e->pb.dval, blk0, e->pb.inv, cplx ? 1 : 0); // the codes stay 0 (+0.0)
	} else if (Pf.nloc > 0)
		k_pb_diag_codes<ASM_HUBBARD><<<nb, kBlock, 0, st>>>(Pf, e->pb.pitch, e->pb.dict, e->pb.ndict, e->pb.dcode, blk0, e->pb.inv, cplx ? 1 : 0);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	*done = true;
	return LPP_OK;
}

} // namespace

namespace {
// the term list and couplings of the S = 1/2 Heisenberg model on the device (lpp_engine_assemble_heisenberg, assemble_heisenberg_raw, pb_chain)
struct HeisDev {
	DevBuf procs, comb, f, z;
};
lpp_status heis_asm_params(lpp_engine* e, int L, int szPlusConst, const double* jpm, const double* jzz, const double* field, int nfield, HeisDev& D, AsmParams& P)
{
	const std::vector<uint64_t> comb = comb_table();
	const int64_t nrows = (int64_t)binom(comb, L, szPlusConst);
	// terms: raise site i (0->1), lower site j (1->0) for every ordered pair with jpm_(i,j) != 0,
	// value 0.5*sqrt(..)*sqrt(..)*jpm = 0.5*jpm for S=1/2   (Heisenberg.h:278-307)
	std::vector<HostProc> hp;
	for (int i = 0; i < L; i++)
		for (int j = 0; j < L; j++) {
			if (i == j || jpm[i * L + j] == 0) continue;
			push(hp, bit(j), bit(i), bit(i) | bit(j), 0, 0, 0, 0.5 * 1.0 * jpm[i * L + j], 0.0, true);
		}
	std::vector<Proc> procs;
	int nneg = 0;
	lpp_status st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;
	if ((st = upload(e->stream, D.procs, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, D.comb, comb.data(), sizeof(uint64_t) * comb.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, D.f, field, sizeof(double) * (size_t)std::max(nfield, 0))) != LPP_OK) return st;
	if ((st = upload(e->stream, D.z, jzz, sizeof(double) * L * L)) != LPP_OK) return st;
	HIP_TRY(hipStreamSynchronize(e->stream)); // the host copies above are locals
	P = AsmParams {};
	P.model = ASM_HEISENBERG;
	P.L = L;
	P.nup = szPlusConst;
	P.ndown = 0;
	P.nproc = (int)procs.size();
	P.nneg = nneg;
	P.n_up = nrows;
	P.nrows_global = nrows;
	P.procs = (const Proc*)D.procs.p;
	P.comb = (const uint64_t*)D.comb.p;
	P.d0 = (const double*)D.f.p;
	P.nd0 = std::min<int>(std::max(nfield, 0), L);
	P.d1 = nullptr;
	P.nd1 = 0;
	P.d2 = (const double*)D.z.p;
	P.row0 = 0;
	P.nloc = nrows;
	P.part = 0;
	return LPP_OK;
}
} // namespace

namespace {
// A chain (couplings between neighbours, and between the two ends): S+S- moves an up spin and nothing sits between the two sites, so the
// off-diagonal part is the hopping matrix of the up spins -- one block of the product-basis form, the in-block kernel decomposed by the high
// sites of the basis word (pb_chain, lpp_pbseg.h).  Amplitudes in the planner's convention: value x (-1)^(up spins between), which for the
// bond between the two ends is the constant (-1)^(n - 1).
lpp_status heis_chain_try(lpp_engine* e, const AsmParams& P, int L, int szPlusConst, const double* jpm, bool* as_chain)
{
	*as_chain = false;
	bool chain = L >= 2 && !e->is_complex;
	std::vector<double> hv((size_t)L * L, 0.0);
	for (int i = 0; i < L && chain; i++)
		for (int j = 0; j < L && chain; j++) {
			if (i == j || jpm[i * L + j] == 0) continue;
			const int d = i > j ? i - j : j - i;
			if ((d != 1 && d != L - 1) || std::memcmp(&jpm[i * L + j], &jpm[j * L + i], sizeof(double)) != 0) chain = false;
			const double v = 0.5 * 1.0 * jpm[i * L + j];
			hv[(size_t)i * L + j] = (d == L - 1 && d != 1 && ((szPlusConst - 1) & 1)) ? -v : v;
		}
	if (!chain || szPlusConst < 1 || szPlusConst >= L) return LPP_OK;
	return pb_chain(e, P, L, szPlusConst, hv, as_chain);
}
// the host copy of the couplings lpp_engine_get_csr re-runs the assembler from
void chain_remember(lpp_engine* e, int L, int m, const double* jpm, const double* jzz, const double* field, int nfield)
{
	PbState& B = e->pb;
	B.chain_L = L;
	B.chain_m = m;
	B.chain_nfield = std::max(nfield, 0);
	B.chain_jpm.assign(jpm, jpm + (size_t)L * L);
	B.chain_jzz.assign(jzz, jzz + (size_t)L * L);
	B.chain_field.clear();
	if (field && nfield > 0) B.chain_field.assign(field, field + nfield);
}
// 1 where two device arrays of `bytes` bytes differ
__global__ void k_bytes_differ(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, size_t bytes, int* __restrict__ flag)
{
	const size_t n8 = bytes / 8;
	bool bad = false;
	for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n8; k += (size_t)gridDim.x * blockDim.x) bad |= ((const uint64_t*)a)[k] != ((const uint64_t*)b)[k];
	if (blockIdx.x == 0 && threadIdx.x == 0)
		for (size_t k = n8 * 8; k < bytes; k++) bad |= a[k] != b[k];
	if (bad) *flag = 1;
}
} // namespace

namespace lpp {
lpp_status assemble_heisenberg_raw(lpp_engine* e, int L, int m, const double* jpm, const double* jzz, const double* field, int nfield, DevCsr& A)
{
	HeisDev D;
	AsmParams P {};
	lpp_status st = heis_asm_params(e, L, m, jpm, jzz, field, nfield, D, P);
	if (st != LPP_OK) return st;
	return dispatch<ASM_HEISENBERG>(e, P, A, 0, 0, true);
}
} // namespace lpp

namespace {
// the term list and couplings of the one-orbital t-J model on the device (lpp_engine_assemble_tj, assemble_tj_raw, tj_build)
struct TjDev {
	DevBuf procs, comb, pv, z, w;
};
lpp_status tj_asm_params(lpp_engine* e, const TjModel& M, TjDev& D, AsmParams& P)
{
	const int L = M.L, nup = M.nup, ndown = M.ndown;
	const std::vector<uint64_t> comb = comb_table();
	const int64_t cfree = (int64_t)binom(comb, L - ndown, nup);
	const int64_t nrows = (int64_t)binom(comb, L, ndown) * cfree;
	const double* hop_re = M.hop_re.data();
	const double* hop_im = M.hop_im.empty() ? nullptr : M.hop_im.data();
	const double* jpm = M.jpm.data();
	std::vector<HostProc> hp;
	for (int i = 0; i < L; i++) {
		for (int j = i + 1; j < L; j++) { // the reference only visits j >= i (TjMultiOrb.h:666,725); i == j never applies
			const double hr = hop_re[i * L + j], hi = hop_im ? hop_im[i * L + j] : 0.0;
			if (hr != 0 || hi != 0) {
				// hopping, TjMultiOrb.h:673-692: value h*extraSign*doSign(ket_s,i,j), doSign = parity of bits [i,j)
				for (int spin = 0; spin < 2; spin++) {
					const int sh = spin * L, oh = (1 - spin) * L; // own / other species shift
					// s_i=1, s_j=0: needs the other species absent at j; extraSign = -1
					push(hp, bit(i + sh), bit(j + sh) | bit(j + oh), bit(i + sh) | bit(j + sh), range_mask(i, j) << sh, 0, 1, hr, hi);
					// s_i=0, s_j=1: needs the other species absent at i; extraSign = +1
					push(hp, bit(j + sh), bit(i + sh) | bit(i + oh), bit(i + sh) | bit(j + sh), range_mask(i, j) << sh, 0, 0, hr, hi);
				}
			}
			const double h = jpm[i * L + j] * 0.5; // TjMultiOrb.h:736
			if (h != 0) {
				const uint64_t x4 = bit(i) | bit(j) | bit(i + L) | bit(j + L);
				const uint64_t sm = range_mask(i, j) | (range_mask(i, j) << L); // signSplusSminus on bra1,bra2 (:772-783)
				// up at i, down at j  ->  up at j, down at i   (:743-754)
				push(hp, bit(i) | bit(j + L), bit(j) | bit(i + L), x4, 0, sm, 0, h, 0.0, true);
				// up at j, down at i  ->  up at i, down at j   (:756-767)
				push(hp, bit(j) | bit(i + L), bit(i) | bit(j + L), x4, 0, sm, 0, h, 0.0, true);
			}
		}
	}
	std::vector<Proc> procs;
	int nneg = 0;
	lpp_status st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;
	std::vector<double> pv(2 * (size_t)L, 0.0);
	const int npv = std::min<int>(std::max(M.npot, 0), L);
	if (M.has_pv)
		for (int i = 0; i < 2 * L; i++) pv[i] = M.pv[(size_t)i];
	if ((st = upload(e->stream, D.procs, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, D.comb, comb.data(), sizeof(uint64_t) * comb.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, D.pv, pv.data(), sizeof(double) * pv.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, D.z, M.jzz.data(), sizeof(double) * L * L)) != LPP_OK) return st;
	if ((st = upload(e->stream, D.w, M.w.data(), sizeof(double) * L * L)) != LPP_OK) return st;
	HIP_TRY(hipStreamSynchronize(e->stream)); // the host copies above are locals
	P = AsmParams {};
	P.model = ASM_TJ;
	P.L = L;
	P.nup = nup;
	P.ndown = ndown;
	P.nproc = (int)procs.size();
	P.nneg = nneg;
	P.n_up = cfree;
	P.nrows_global = nrows;
	P.procs = (const Proc*)D.procs.p;
	P.comb = (const uint64_t*)D.comb.p;
	P.d0 = (const double*)D.pv.p;
	P.nd0 = M.has_pv ? npv : 0;
	P.d1 = (const double*)D.z.p;
	P.d2 = (const double*)D.w.p;
	P.row0 = 0;
	P.nloc = nrows;
	P.part = 0;
	return LPP_OK;
}
} // namespace

namespace lpp {
// the plain CSR of the model in the reference's order, nothing else (lpp_engine_get_csr of the hole-major form)
lpp_status assemble_tj_raw(lpp_engine* e, const TjModel& M, DevCsr& A)
{
	TjDev D;
	AsmParams P {};
	lpp_status st = tj_asm_params(e, M, D, P);
	if (st != LPP_OK) return st;
	return dispatch<ASM_TJ>(e, P, A, 0, 0, true);
}
} // namespace lpp

namespace lpp {
lpp_status model_layout_from_hint(lpp_engine* e, const DevCsr& A, bool* done)
{
	*done = false;
	const ModelHint& H = e->hint;
	if (H.kind == 0 || e->has_comm || A.nrows <= 0 || !A.rowptr || !A.col || !A.val) return LPP_OK;
	const std::vector<uint64_t> comb = comb_table();
	const bool verbose = getenv("LPP_VERBOSE") != nullptr;
	hipStream_t st = e->stream;
	// (1) the description must regenerate the handed-over matrix, bit for bit
	DevCsr T;
	struct Drop {
		DevCsr& t;
		~Drop() { free_csr(t); }
	} drop { T };
	lpp_status rc = LPP_OK;
	if (H.kind == 1) {
		const TjModel& M = H.tj;
		if ((int64_t)binom(comb, M.L, M.ndown) * (int64_t)binom(comb, M.L - M.ndown, M.nup) != A.nrows || (M.has_im && !e->is_complex)) return LPP_OK;
		if (!tj_applies(e, M)) return LPP_OK; // (no regeneration for a form that would not be taken anyway)
		rc = assemble_tj_raw(e, M, T);
	} else if (H.kind == 2) {
		if ((int64_t)binom(comb, H.L, H.m) != A.nrows || e->is_complex) return LPP_OK;
		if ((size_t)A.nrows * sizeof(double) < ((size_t)32 << 20) && !getenv("LPP_PRODUCT_LAYOUT")) return LPP_OK; // (pb_chain's own threshold)
		rc = assemble_heisenberg_raw(e, H.L, H.m, H.jpm.data(), H.jzz.data(), H.field.empty() ? nullptr : H.field.data(), H.nfield, T);
	} else
		return LPP_OK;
	if (rc == LPP_ERR_NOMEM || rc == LPP_ERR_INVALID) return LPP_OK; // no room for the second copy / a description the assembler refuses: the general layout
	if (rc != LPP_OK) return rc;
	bool same = T.nrows == A.nrows && T.nnz == A.nnz;
	if (same) {
		DevBuf flag;
		HIP_TRY_MEM(hipMalloc(&flag.p, sizeof(int)));
		HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), st));
		k_bytes_differ<<<2048, 256, 0, st>>>((const unsigned char*)T.rowptr, (const unsigned char*)A.rowptr, sizeof(int64_t) * (size_t)(A.nrows + 1), (int*)flag.p);
		k_bytes_differ<<<2048, 256, 0, st>>>((const unsigned char*)T.col, (const unsigned char*)A.col, sizeof(int32_t) * (size_t)A.nnz, (int*)flag.p);
		k_bytes_differ<<<2048, 256, 0, st>>>((const unsigned char*)T.val, (const unsigned char*)A.val, e->esz * (size_t)A.nnz, (int*)flag.p);
		int bad = 0;
		HIP_TRY(hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(st));
		same = bad == 0;
	}
	free_csr(T);
	if (verbose) fprintf(stderr, "lpp: the model description %s the handed-over CSR\n", same ? "regenerates" : "does NOT regenerate");
	if (!same) return LPP_OK;
	// (2) the structured form of that model (checked on its own against the assembler's row walk)
	if (H.kind == 1) {
		TjDev D;
		AsmParams P {};
		if ((rc = tj_asm_params(e, H.tj, D, P)) != LPP_OK) return rc;
		return tj_build(e, H.tj, P, done);
	}
	HeisDev D;
	AsmParams P {};
	if ((rc = heis_asm_params(e, H.L, H.m, H.jpm.data(), H.jzz.data(), H.field.empty() ? nullptr : H.field.data(), H.nfield, D, P)) != LPP_OK) return rc;
	if ((rc = heis_chain_try(e, P, H.L, H.m, H.jpm.data(), done)) != LPP_OK) return rc;
	if (*done) chain_remember(e, H.L, H.m, H.jpm.data(), H.jzz.data(), H.field.empty() ? nullptr : H.field.data(), H.nfield);
	return LPP_OK;
}
} // namespace lpp

extern "C" {

lpp_status lpp_engine_assemble_hubbard(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                       const double* hop_re, const double* hop_im, const double* U, const double* V)
{
	return lpp_engine_assemble_hubbard_ext(e, comm, L, nup, ndown, hop_re, hop_im, U, V, nullptr);
}

lpp_status lpp_engine_assemble_hubbard_ext(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                           const double* hop_re, const double* hop_im, const double* U, const double* V,
                                           const double* ninj)
{
	return lpp_engine_assemble_hubbard_super(e, comm, L, nup, ndown, hop_re, hop_im, U, V, ninj, nullptr);
}

lpp_status lpp_engine_assemble_hubbard_super(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                             const double* hop_re, const double* hop_im, const double* U, const double* V,
                                             const double* ninj, const double* jcoup)
{
	if (!e || !hop_re || !U || !V || L < 1 || L > 31 || nup < 0 || ndown < 0 || nup > L || ndown > L)
		return fail(LPP_ERR_INVALID, "lpp_engine_assemble_hubbard: bad argument (1 <= L <= 31)");
	const std::vector<uint64_t> comb = comb_table();
	const int64_t n_up = (int64_t)binom(comb, L, nup), n_dn = (int64_t)binom(comb, L, ndown);
	const int64_t nrows = n_up * n_dn;
	bool cplx_in = false;
	if (hop_im)
		for (int k = 0; k < L * L; k++) cplx_in |= (hop_im[k] != 0);
	// with the transposition exchange a rank stores columns of its own slice and of its transposed slice only
	const bool local_columns = comm && comm->nranks > 1 && comm->exchange_begin && comm->exchange_end && comm->xchg_chunk > 0;
	lpp_status st = common_setup(e, nrows, cplx_in, local_columns);
	if (st != LPP_OK) return st;

	bool has_j = false; // Model=SuperHubbardExtended: spin coupling (geometry term 2)
	if (jcoup)
		for (int k = 0; k < L * L; k++) has_j |= (jcoup[k] != 0);
	if (!has_j) jcoup = nullptr;
	std::vector<HostProc> hp;
	hubbard_terms(L, hop_re, hop_im, hp);
	if (jcoup) super_terms(L, jcoup, hp);
	std::vector<Proc> procs;
	int nneg = 0;
	st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;

	DevBuf d_procs, d_comb, d_U, d_V, d_U0, d_jc;
	if (jcoup && (st = upload(e->stream, d_jc, jcoup, sizeof(double) * L * L)) != LPP_OK) return st;
	const std::vector<double> zeroU((size_t)L, 0.0);
	if ((st = upload(e->stream, d_procs, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_comb, comb.data(), sizeof(uint64_t) * comb.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_U, U, sizeof(double) * L)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_V, V, sizeof(double) * L)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_U0, zeroU.data(), sizeof(double) * L)) != LPP_OK) return st;
	DevBuf d_ninj;
	if (ninj && (st = upload(e->stream, d_ninj, ninj, sizeof(double) * L * L)) != LPP_OK) return st;

	AsmParams P {};
	P.model = ASM_HUBBARD;
	P.L = L;
	P.nup = nup;
	P.ndown = ndown;
	P.nproc = (int)procs.size();
	P.nneg = nneg;
	P.n_up = n_up;
	P.nrows_global = nrows;
	P.procs = (const Proc*)d_procs.p;
	P.comb = (const uint64_t*)d_comb.p;
	P.d0 = (const double*)d_U.p;
	P.d1 = (const double*)d_V.p;
	P.d2 = ninj ? (const double*)d_ninj.p : nullptr; // Coulomb coupling of HubbardOneBandExtended
	P.d3 = jcoup ? (const double*)d_jc.p : nullptr; // spin coupling of SuperHubbardExtended

	const bool multi = comm && comm->nranks > 1;
	if (!multi) {
		e->has_comm = false;
		e->bind_scalars(e->scal_own);
		free_csr(e->A_rem);
		drop_product(e);
		P.row0 = 0;
		P.nloc = nrows;
		P.part = 0;
		bool as_product = false;
		st = assemble_hubbard_pb(e, P, nup, ndown, n_up, n_dn, (const double*)d_U0.p, &as_product, 0, -1, 0, 0, true);
		if (st != LPP_OK) return st;
		if (as_product) {
			free_csr(e->A_loc);
		} else {
			st = dispatch<ASM_HUBBARD>(e, P, e->A_loc);
			if (st != LPP_OK) return st;
		}
		e->n_local = e->n_global = nrows;
		e->row_start = 0;
	} else {
		st = e->adopt_comm(comm);
		if (st != LPP_OK) return st;
		// partition at multiples of N_up: whole down-configurations per rank, so the diagonal and all
		// up-hops stay rank-local; gathered index == global index because stride == per*N_up
		std::vector<int64_t> starts(comm->nranks + 1);
		st = lpp_partition_rows(nrows, comm->nranks, n_up, starts.data());
		if (st != LPP_OK) return st;
		const int64_t per = (n_dn + comm->nranks - 1) / comm->nranks;
		if (comm->shard_stride != per * n_up) return fail(LPP_ERR_INVALID, "assemble_hubbard: comm.shard_stride must be ceil(N_down/nranks)*N_up");
		const bool transpose = comm->exchange_begin && comm->exchange_end && comm->xchg_chunk > 0;
		if (transpose && jcoup) return fail(LPP_ERR_INVALID, "assemble_hubbard: spin-flip terms (SuperHubbardExtended) need the all-gather exchange");
		// all-gather: remote columns index the gathered vector; transposition: columns index the rank's own slice (and its
		// transposed slice, checked below), so the GLOBAL dimension may exceed 2^31 (e.g. the 3.0e9-state (7,6) sector of the 4x5 lattice)
		if (!transpose && (int64_t)comm->nranks * comm->shard_stride > (int64_t)INT32_MAX)
			return fail(LPP_ERR_INVALID, "assemble_hubbard: gathered vector exceeds 32-bit column range (use the transposition exchange)");
		if (comm->shard_stride > (int64_t)INT32_MAX) return fail(LPP_ERR_INVALID, "assemble_hubbard: a rank's slice exceeds 32-bit column range");
		P.row0 = starts[comm->rank];
		P.nloc = starts[comm->rank + 1] - starts[comm->rank];
		P.col_lo = starts[comm->rank];
		P.col_hi = starts[comm->rank + 1];
		drop_product(e);
		if (!transpose) {
			e->tx = false;
			P.part = 1;
			st = dispatch<ASM_HUBBARD>(e, P, e->A_loc);
			if (st != LPP_OK) return st;
			P.part = 2;
			e->A_rem.src_elems = (int64_t)comm->nranks * comm->shard_stride;
			st = dispatch<ASM_HUBBARD>(e, P, e->A_rem);
			if (st != LPP_OK) return st;
		} else {
			// Transposition scheme: the up-hop + diagonal part acts on the rank's own slice (down-index partition);
			// the down-hop part is assembled for the rank's UP-index range over ALL down indices, in the layout
			// row = id*peru + (iu - iu0), and acts on the transposed slice delivered by the first all-to-all.
			// xchg_chunk = per * peru with peru >= ceil(N_up/P) up indices per rank; a caller that rounds peru up to a multiple of 16
			// gets the product-basis kernels on both parts when the matrix qualifies
			const int64_t peru = per > 0 ? comm->xchg_chunk / per : 0;
			if (!comm->send2_buf || !comm->recv2_buf || per <= 0 || comm->xchg_chunk != per * peru || peru * comm->nranks < n_up)
				return fail(LPP_ERR_INVALID, "assemble_hubbard: transposition exchange needs send2/recv2 buffers and xchg_chunk == ceil(N_down/P) * peru, peru >= ceil(N_up/P)");
			if ((int64_t)comm->nranks * per * peru > (int64_t)INT32_MAX) return fail(LPP_ERR_INVALID, "assemble_hubbard: transposed slice exceeds 32-bit column range");
			if ((peru & 15) == 0 && P.nloc % n_up == 0) {
				bool as_product = false;
				st = assemble_hubbard_pb(e, P, nup, ndown, n_up, n_dn, (const double*)d_U0.p, &as_product, P.row0 / n_up, P.nloc / n_up, peru, (int64_t)comm->nranks * per);
				if (st != LPP_OK) return st;
				if (as_product) {
					free_csr(e->A_loc);
					free_csr(e->A_rem);
					e->tx = true;
					e->tx_per = per;
					e->tx_peru = peru;
					e->kron_n_up_tx = n_up;
					e->n_local = P.nloc;
					e->n_global = nrows;
					e->row_start = P.row0;
					e->active = false;
					set_spmv_bytes(e);
					return alloc_work(e);
				}
			}
			std::vector<HostProc> hu, hd;
			hubbard_terms(L, hop_re, hop_im, hu, 1);
			hubbard_terms(L, hop_re, hop_im, hd, 2);
			std::vector<Proc> pu, pd;
			int nnegu = 0, nnegd = 0;
			if ((st = finish_procs(hu, pu, &nnegu)) != LPP_OK) return st;
			if ((st = finish_procs(hd, pd, &nnegd)) != LPP_OK) return st;
			DevBuf d_pu, d_pd;
			if ((st = upload(e->stream, d_pu, pu.data(), sizeof(Proc) * pu.size())) != LPP_OK) return st;
			if ((st = upload(e->stream, d_pd, pd.data(), sizeof(Proc) * pd.size())) != LPP_OK) return st;
			AsmParams Pu = P;
			Pu.procs = (const Proc*)d_pu.p;
			Pu.nproc = (int)pu.size();
			Pu.nneg = nnegu;
			Pu.part = 1;
			st = dispatch<ASM_HUBBARD>(e, Pu, e->A_loc);
			if (st != LPP_OK) return st;
			AsmParams Pd = P;
			Pd.procs = (const Proc*)d_pd.p;
			Pd.nproc = (int)pd.size();
			Pd.nneg = nnegd;
			Pd.part = 0;
			Pd.no_diag = 1;
			Pd.tr = 1;
			Pd.peru = peru;
			Pd.iu0 = std::min<int64_t>((int64_t)comm->rank * peru, n_up);
			Pd.nu = std::min<int64_t>(Pd.iu0 + peru, n_up) - Pd.iu0;
			Pd.n_dn = n_dn;
			Pd.row0 = 0;
			Pd.nloc = (int64_t)comm->nranks * per * peru;
			e->A_rem.src_elems = Pd.nloc;
			st = dispatch<ASM_HUBBARD>(e, Pd, e->A_rem, LPP_SPMV_SLICED, peru);
			if (st != LPP_OK) return st;
			e->A_rem.hint_block = 0;
			e->tx = true;
			e->tx_per = per;
			e->tx_peru = peru;
			e->kron_n_up_tx = n_up;
		}
		e->n_local = P.nloc;
		e->n_global = nrows;
		e->row_start = P.row0;
	}
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

lpp_status lpp_engine_assemble_heisenberg(lpp_engine* e, int32_t L, int32_t szPlusConst, const double* jpm, const double* jzz,
                                          const double* field, int32_t nfield)
{
	if (!e || !jpm || !jzz || L < 1 || L > 62 || szPlusConst < 0 || szPlusConst > L)
		return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg: bad argument (S=1/2, 1 <= L <= 62)");
	const std::vector<uint64_t> comb = comb_table();
	const int64_t nrows = (int64_t)binom(comb, L, szPlusConst);
	lpp_status st = common_setup(e, nrows, 0);
	if (st != LPP_OK) return st;
	HeisDev D;
	AsmParams P {};
	if ((st = heis_asm_params(e, L, szPlusConst, jpm, jzz, field, nfield, D, P)) != LPP_OK) return st;
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_rem);
	drop_product(e);
	// A chain (couplings between neighbours, and between the two ends): one block of the product-basis form, planned from the couplings alone
	// (heis_chain_try): no CSR is assembled (lpp_engine_get_csr re-runs the assembler), so the largest chain is bounded by its vectors
	bool as_chain = false;
	if ((st = heis_chain_try(e, P, L, szPlusConst, jpm, &as_chain)) != LPP_OK) return st;
	if (as_chain) {
		free_csr(e->A_loc);
		chain_remember(e, L, szPlusConst, jpm, jzz, field, nfield);
	} else {
		st = dispatch<ASM_HEISENBERG>(e, P, e->A_loc);
		if (st != LPP_OK) return st;
	}
	e->n_local = e->n_global = nrows;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

// Heisenberg with any spin the reference's digit width can hold (BasisHeisenberg.h:28-46: bits = 1 + floor(log2(twiceS + 1)), one
// less for odd twiceS -- so odd spins fit only when twiceS + 1 is a power of two), with the anisotropy term (Heisenberg.h:259).
// twiceS == 1 gives the matrix of lpp_engine_assemble_heisenberg through the digit basis (one-bit digits).
// The S+S- value follows the reference to the letter: both square roots are taken of the LOWERED site's m (Heisenberg.h:296-303),
// which is a constant for S <= 1 and makes the matrix non-symmetric from S = 3/2 on -- reproduced, not repaired.
lpp_status lpp_engine_assemble_heisenberg_spin(lpp_engine* e, int32_t L, int32_t twiceS, int32_t szPlusConst, const double* jpm, const double* jzz,
                                               const double* field, int32_t nfield, const double* anisotropy, int32_t naniso)
{
	if (!e || !jpm || !jzz || L < 1 || twiceS < 1 || szPlusConst < 0) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg_spin: bad argument");
	nfield = std::max(nfield, 0);
	naniso = std::max(naniso, 0);
	if ((nfield > 0 && !field) || (naniso > 0 && !anisotropy)) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg_spin: null field / anisotropy");
	int bits = 1;
	while ((2 << (bits - 1)) <= twiceS + 1) bits++; // 1 + floor(log2(twiceS + 1))
	if (twiceS & 1) bits--;
	const int dmax = (twiceS & 1) ? (1 << bits) - 1 : twiceS; // even spins: digits above twiceS are filtered out (mOf, :204-227)
	if (dmax < twiceS) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg_spin: the reference's digit width cannot hold this spin (odd twiceS with twiceS + 1 not a power of two)");
	if ((int64_t)bits * L > 62) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg_spin: bits * L > 62");
	if (szPlusConst > dmax * L) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_heisenberg_spin: empty Hilbert space");
	// digits[l][s]: l-digit strings with digit sum s
	const int sdim = dmax * L + 1;
	std::vector<uint64_t> dig((size_t)(L + 1) * sdim, 0);
	dig[0] = 1;
	for (int l = 1; l <= L; l++)
		for (int sum = 0; sum < sdim; sum++) {
			uint64_t c = 0;
			for (int d = 0; d <= dmax && d <= sum; d++) c += dig[(size_t)(l - 1) * sdim + (sum - d)];
			dig[(size_t)l * sdim + sum] = c;
		}
	const int64_t nrows = (int64_t)dig[(size_t)L * sdim + szPlusConst];
	lpp_status st = common_setup(e, nrows, 0);
	if (st != LPP_OK) return st;
	// terms: raise digit i, lower digit j for every ordered pair with jpm_(i,j) != 0 (Heisenberg.h:101-106, 290-306); the sites ride in
	// smask_ket / smask_bra until the list is sorted (they are not used as masks by this model)
	std::vector<HostProc> hp;
	for (int i = 0; i < L; i++)
		for (int j = 0; j < L; j++) {
			if (i == j || jpm[i * L + j] == 0) continue;
			HostProc h {};
			h.p.need_set = (uint64_t)(i * bits);
			h.p.need_clear = (uint64_t)(j * bits);
			h.p.smask_ket = (uint64_t)i;
			h.p.smask_bra = (uint64_t)j;
			h.p.real_only = 1;
			h.delta = (int64_t)(1ull << (i * bits)) - (int64_t)(1ull << (j * bits));
			hp.push_back(h);
		}
	std::vector<Proc> procs;
	int nneg = 0;
	st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;
	// value of term (i, j) on a ket whose digit j is v: the reference's expression, operation by operation (Heisenberg.h:296-304)
	const double spin = twiceS * 0.5;
	std::vector<double> amp(std::max<size_t>(procs.size() * (size_t)(twiceS + 1), 1), 0.0);
	for (size_t p = 0; p < procs.size(); p++) {
		const int i = (int)procs[p].smask_ket, j = (int)procs[p].smask_bra;
		for (int v = 1; v <= twiceS; v++) {
			int val2 = v;
			const double m2 = val2 - spin;
			val2--;
			const double m1 = val2 - spin;
			double tmp = std::sqrt(spin * (spin + 1.0) - m1 * (m1 + 1.0));
			tmp *= std::sqrt(spin * (spin + 1.0) - m2 * (m2 - 1.0));
			amp[p * (size_t)(twiceS + 1) + v] = 0.5 * tmp * jpm[i * L + j];
		}
		procs[p].smask_ket = procs[p].smask_bra = 0;
	}
	DevBuf d_procs, d_dig, d_f, d_a, d_z, d_amp;
	if ((st = upload(e->stream, d_procs, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_dig, dig.data(), sizeof(uint64_t) * dig.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_f, field, sizeof(double) * (size_t)nfield)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_a, anisotropy, sizeof(double) * (size_t)naniso)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_z, jzz, sizeof(double) * L * L)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_amp, amp.data(), sizeof(double) * amp.size())) != LPP_OK) return st;
	AsmParams P {};
	P.model = ASM_HEISENBERG_S;
	P.L = L;
	P.nproc = (int)procs.size();
	P.nneg = nneg;
	P.n_up = nrows;
	P.nrows_global = nrows;
	P.procs = (const Proc*)d_procs.p;
	P.d0 = (const double*)d_f.p;
	P.nd0 = std::min<int>(nfield, L);
	P.d1 = (const double*)d_a.p;
	P.nd1 = std::min<int>(naniso, L);
	P.d2 = (const double*)d_z.p;
	P.twiceS = twiceS;
	P.bits = bits;
	P.dmax = dmax;
	P.msum = szPlusConst;
	P.sdim = sdim;
	P.digits = (const uint64_t*)d_dig.p;
	P.amp = (const double*)d_amp.p;
	P.row0 = 0;
	P.nloc = nrows;
	P.part = 0;
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_rem);
	drop_product(e);
	st = dispatch<ASM_HEISENBERG_S>(e, P, e->A_loc);
	if (st != LPP_OK) return st;
	e->n_local = e->n_global = nrows;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

lpp_status lpp_engine_assemble_tj(lpp_engine* e, int32_t L, int32_t nup, int32_t ndown, const double* hop_re, const double* hop_im,
                                  const double* jpm, const double* jzz, const double* w, const double* potentialV, int32_t npot)
{
	if (!e || !hop_re || !jpm || !jzz || !w || L < 1 || L > 31 || nup < 0 || ndown < 0 || nup + ndown > L)
		return fail(LPP_ERR_INVALID, "lpp_engine_assemble_tj: bad argument (1 <= L <= 31, nup+ndown <= L)");
	if (potentialV && npot > 0 && npot < 2 * L) return fail(LPP_ERR_INVALID, "lpp_engine_assemble_tj: potentialV needs 2*L entries (up then down)");
	TjModel M;
	M.L = L;
	M.nup = nup;
	M.ndown = ndown;
	M.npot = npot;
	const size_t LL = (size_t)L * L;
	M.hop_re.assign(hop_re, hop_re + LL);
	if (hop_im) M.hop_im.assign(hop_im, hop_im + LL);
	M.jpm.assign(jpm, jpm + LL);
	M.jzz.assign(jzz, jzz + LL);
	M.w.assign(w, w + LL);
	M.has_pv = potentialV && npot > 0;
	if (M.has_pv) M.pv.assign(potentialV, potentialV + 2 * (size_t)L);
	if (hop_im)
		for (size_t k = 0; k < LL; k++) M.has_im |= (hop_im[k] != 0);
	const std::vector<uint64_t> comb = comb_table();
	const int64_t nrows = (int64_t)binom(comb, L, ndown) * (int64_t)binom(comb, L - ndown, nup);
	lpp_status st = common_setup(e, nrows, M.has_im);
	if (st != LPP_OK) return st;
	TjDev D;
	AsmParams P {};
	if ((st = tj_asm_params(e, M, D, P)) != LPP_OK) return st;
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_rem);
	drop_product(e);
	// the hole-major form without a stored matrix (lpp_tj_kernels.h) where it applies; otherwise the CSR in the general layout
	bool as_tj = false;
	if ((st = tj_build(e, M, P, &as_tj)) != LPP_OK) return st;
	if (as_tj)
		free_csr(e->A_loc);
	else {
		st = dispatch<ASM_TJ>(e, P, e->A_loc);
		if (st != LPP_OK) return st;
	}
	e->n_local = e->n_global = nrows;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}


} // extern "C"

namespace {
// H_up on the device in the layout the product kernel wants: packed (col16|code|code) when it qualifies,
// otherwise the generic sliced layout.
template <typename T> lpp_status build_kron_up(lpp_engine* e, int64_t n_up)
{
	KronState& K = e->kron;
	DevCsr& A = K.up;
	const int ncomp = (int)(sizeof(T) / sizeof(double));
	bool packed_ok = n_up <= (ncomp == 2 ? 65536 : (1 << 24)) && getenv("LPP_KRON_NO_PACK") == nullptr;
	std::vector<int64_t> rp;
	std::vector<int32_t> ci;
	std::vector<double> va;
	std::vector<double> dict;
	if (packed_ok) {
		rp.resize(n_up + 1);
		ci.resize(std::max<int64_t>(A.nnz, 1));
		va.resize((size_t)std::max<int64_t>(A.nnz, 1) * ncomp);
		HIP_TRY(hipMemcpy(rp.data(), A.rowptr, sizeof(int64_t) * (size_t)(n_up + 1), hipMemcpyDeviceToHost));
		if (A.nnz) {
			HIP_TRY(hipMemcpy(ci.data(), A.col, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(va.data(), A.val, sizeof(T) * (size_t)A.nnz, hipMemcpyDeviceToHost));
		}
		dict.push_back(0.0);
		for (size_t k = 0; k < (size_t)A.nnz * ncomp && dict.size() <= 256; k++) {
			bool found = false;
			for (double d : dict)
				if (std::memcmp(&d, &va[k], sizeof(double)) == 0) {
					found = true;
					break;
				}
			if (!found) dict.push_back(va[k]);
		}
		if (dict.size() > 256) packed_ok = false;
	}
	if (!packed_ok) {
		K.packed = false;
		A.no_dia = true; // the matrix-free kernel walks the plain sliced layout
		return finalize_csr(e, A, true, LPP_SPMV_SLICED, n_up);
	}
	auto code_of = [&](double v) -> uint32_t {
		for (size_t i = 0; i < dict.size(); i++)
			if (std::memcmp(&dict[i], &v, sizeof(double)) == 0) return (uint32_t)i;
		return 0;
	};
	const int spb = (int)((n_up + 63) / 64);
	// one packing per LDS window piece: the whole row when N_up fits LDS (nchunk = 1), otherwise nchunk pieces of cw
	// columns, each holding the entries whose column lies in the piece, with columns relative to it
	const int64_t cap = (int64_t)((156 * 1024) / sizeof(T));
	const int nchunk = (int)std::max<int64_t>(1, (n_up + cap - 1) / cap);
	const int64_t cw = nchunk == 1 ? std::max<int64_t>(n_up, 64) : (((n_up + nchunk - 1) / nchunk + 63) / 64) * 64;
	std::vector<int32_t> off((size_t)nchunk * (spb + 1), 0), len((size_t)nchunk * spb, 0);
	std::vector<uint32_t> words;
	std::vector<int64_t> cnt(64);
	for (int c = 0; c < nchunk; c++) {
		const int64_t c0 = (int64_t)c * cw, c1 = std::min<int64_t>(c0 + cw, n_up);
		int32_t* offc = off.data() + (size_t)c * (spb + 1);
		int32_t* lenc = len.data() + (size_t)c * spb;
		for (int s2 = 0; s2 < spb; s2++) {
			int64_t mx = 0;
			for (int64_t r = (int64_t)s2 * 64; r < std::min<int64_t>(n_up, (int64_t)(s2 + 1) * 64); r++) {
				int64_t n = 0;
				for (int64_t p = rp[r]; p < rp[r + 1]; p++) n += (ci[p] >= c0 && ci[p] < c1) ? 1 : 0;
				mx = std::max(mx, n);
			}
			lenc[s2] = (int32_t)((mx + 7) / 8 * 8);
			offc[s2] = (int32_t)words.size();
			words.resize(words.size() + (size_t)lenc[s2] * 64, 0u); // padding: column 0 of the piece, value 0.0 (code 0)
			for (int lane = 0; lane < 64; lane++) {
				const int64_t r = (int64_t)s2 * 64 + lane;
				if (r >= n_up) continue;
				int k = 0;
				for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
					if (ci[p] < c0 || ci[p] >= c1) continue;
					uint32_t w;
					if (ncomp == 2)
						w = (uint32_t)(ci[p] - c0) | (code_of(va[(size_t)p * 2]) << 16) | (code_of(va[(size_t)p * 2 + 1]) << 24);
					else
						w = (uint32_t)(ci[p] - c0) | (code_of(va[p]) << 24);
					words[(size_t)offc[s2] + (size_t)k * 64 + lane] = w;
					k++;
				}
			}
		}
		offc[spb] = (int32_t)words.size();
		if (words.size() > ((size_t)1 << 30)) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: packed H_up too large");
	}
	words.resize(words.size() + 64, 0u);
	K.pk_nchunk = nchunk;
	K.pk_cw = (int)cw;
	dict.resize(256, 0.0);
	HIP_TRY_MEM(hipMalloc(&K.pk_words, sizeof(uint32_t) * words.size()));
	HIP_TRY_MEM(hipMalloc(&K.pk_off, sizeof(int32_t) * off.size()));
	HIP_TRY_MEM(hipMalloc(&K.pk_len, sizeof(int32_t) * std::max<size_t>(len.size(), 1)));
	HIP_TRY_MEM(hipMalloc(&K.pk_dict, sizeof(double) * 256));
	HIP_TRY(hipMemcpy(K.pk_words, words.data(), sizeof(uint32_t) * words.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(K.pk_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(K.pk_len, len.data(), sizeof(int32_t) * len.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(K.pk_dict, dict.data(), sizeof(double) * 256, hipMemcpyHostToDevice));
	K.pk_spb = spb;
	K.packed = true;
	return LPP_OK;
}
} // namespace

extern "C" {

lpp_status lpp_engine_setup_hubbard_onthefly(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                             const double* hop_re, const double* hop_im, const double* U, const double* V)
{
	return lpp_engine_setup_hubbard_onthefly_ext(e, comm, L, nup, ndown, hop_re, hop_im, U, V, nullptr);
}

lpp_status lpp_engine_setup_hubbard_onthefly_ext(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                                 const double* hop_re, const double* hop_im, const double* U, const double* V,
                                                 const double* ninj)
{
	if (!e || !hop_re || !U || !V || L < 1 || L > 31 || nup < 0 || ndown < 0 || nup > L || ndown > L)
		return fail(LPP_ERR_INVALID, "lpp_engine_setup_hubbard_onthefly: bad argument (1 <= L <= 31)");
	const std::vector<uint64_t> comb = comb_table();
	const int64_t n_up = (int64_t)binom(comb, L, nup), n_dn = (int64_t)binom(comb, L, ndown);
	bool cplx_in = false;
	if (hop_im)
		for (int k = 0; k < L * L; k++) cplx_in |= (hop_im[k] != 0);
	if (cplx_in && !e->is_complex) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: complex hoppings need a c128 engine");
	if (n_up <= 0 || n_dn <= 0) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: empty Hilbert space");
	if (n_up > (int64_t)INT32_MAX || n_dn > (int64_t)INT32_MAX) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: one-species space too large");
	HIP_TRY(hipSetDevice(e->cfg.device));
	const bool multi = comm && comm->nranks > 1;
	lpp_status st = LPP_OK;
	int64_t id0 = 0, nid = n_dn;
	if (multi) {
		st = e->adopt_comm(comm);
		if (st != LPP_OK) return st;
		const int64_t per = (n_dn + comm->nranks - 1) / comm->nranks;
		if (comm->shard_stride != per * n_up) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: comm.shard_stride must be ceil(N_down/nranks)*N_up");
		id0 = std::min<int64_t>((int64_t)comm->rank * per, n_dn);
		nid = std::min<int64_t>(id0 + per, n_dn) - id0;
	} else {
		e->has_comm = false;
		e->bind_scalars(e->scal_own);
	}
	free_csr(e->A_loc);
	free_csr(e->A_rem);
	drop_product(e);
	KronState& K = e->kron;

	std::vector<HostProc> hp;
	hubbard_terms(L, hop_re, hop_im, hp);
	std::vector<Proc> procs;
	int nneg = 0;
	st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;
	std::vector<double> zeroU(L, 0.0);
	DevBuf d_procs, d_comb, d_U0, d_V;
	if ((st = upload(e->stream, d_procs, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_comb, comb.data(), sizeof(uint64_t) * comb.size())) != LPP_OK) return st;
	if ((st = upload(e->stream, d_U0, zeroU.data(), sizeof(double) * L)) != LPP_OK) return st;
	if ((st = upload(e->stream, d_V, V, sizeof(double) * L)) != LPP_OK) return st;
	// Where one species' row fits the LDS window the matrix-free product IS the product-basis one: H = 1 (x) T + C (x) 1 + D with T, C
	// (a few hundred KB) and one diagonal code per row -- an eighth of a vector -- is everything it keeps, and its two kernels move a
	// third of what the fused block-order kernel below moves (10.9 against 31.3 GB per step at BASELINE config 2).  The kernels
	// below serve what does not qualify: rows beyond the window (config 5's sectors), complex hoppings, small problems.
	// LPP_ONTHEFLY_KRON=1 keeps them for everything (tests cross-check the two).
	if (!e->is_complex && !(getenv("LPP_ONTHEFLY_KRON") && atoi(getenv("LPP_ONTHEFLY_KRON")) != 0)) {
		DevBuf d_U, d_nj;
		if ((st = upload(e->stream, d_U, U, sizeof(double) * L)) != LPP_OK) return st;
		if (ninj && (st = upload(e->stream, d_nj, ninj, sizeof(double) * L * L)) != LPP_OK) return st;
		AsmParams Pf {};
		Pf.model = ASM_HUBBARD;
		Pf.L = L;
		Pf.nup = nup;
		Pf.ndown = ndown;
		Pf.nproc = (int)procs.size();
		Pf.nneg = nneg;
		Pf.n_up = n_up;
		Pf.nrows_global = n_up * n_dn;
		Pf.procs = (const Proc*)d_procs.p;
		Pf.comb = (const uint64_t*)d_comb.p;
		Pf.d0 = (const double*)d_U.p;
		Pf.d1 = (const double*)d_V.p;
		Pf.d2 = ninj ? (const double*)d_nj.p : nullptr;
		Pf.row0 = id0 * n_up;
		Pf.nloc = nid * n_up;
		bool as_product = false;
		if (!multi) {
			st = assemble_hubbard_pb(e, Pf, nup, ndown, n_up, n_dn, (const double*)d_U0.p, &as_product);
		} else if (comm->exchange_begin && comm->exchange_end && comm->xchg_chunk > 0 && comm->send2_buf && comm->recv2_buf) {
			const int64_t per = (n_dn + comm->nranks - 1) / comm->nranks, peru = per > 0 ? comm->xchg_chunk / per : 0;
			if (per > 0 && comm->xchg_chunk == per * peru && peru * comm->nranks >= n_up && (peru & 15) == 0) {
				st = assemble_hubbard_pb(e, Pf, nup, ndown, n_up, n_dn, (const double*)d_U0.p, &as_product, id0, nid, peru, (int64_t)comm->nranks * per);
				if (st == LPP_OK && as_product) {
					e->tx = true;
					e->tx_per = per;
					e->tx_peru = peru;
					e->kron_n_up_tx = n_up;
				}
			}
		}
		if (st != LPP_OK) return st;
		if (as_product) {
			e->n_local = nid * n_up;
			e->n_global = n_up * n_dn;
			e->row_start = id0 * n_up;
			e->active = false;
			set_spmv_bytes(e);
			return alloc_work(e);
		}
	}
	// one-species matrices = the Hubbard assembler with the other species empty: hops of that species plus
	// its potential diagonal sum_i V_i n_i (HubbardHelper.h:180-183); the U term is applied by the kernel
	AsmParams P {};
	P.model = ASM_HUBBARD;
	P.L = L;
	P.ndown = 0;
	P.nproc = (int)procs.size();
	P.nneg = nneg;
	P.procs = (const Proc*)d_procs.p;
	P.comb = (const uint64_t*)d_comb.p;
	P.d0 = (const double*)d_U0.p;
	P.d1 = (const double*)d_V.p;
	P.part = 0;
	P.row0 = 0;
	P.nup = nup;
	P.n_up = n_up;
	P.nrows_global = P.nloc = n_up;
	st = dispatch<ASM_HUBBARD>(e, P, K.up, LPP_SPMV_ROWGROUP, 0); // plain CSR first; packed or sliced below
	if (st != LPP_OK) return st;
	P.nup = ndown;
	P.n_up = n_dn;
	P.nrows_global = P.nloc = n_dn;
	st = dispatch<ASM_HUBBARD>(e, P, K.dn, LPP_SPMV_ROWGROUP, 0); // plain CSR
	if (st != LPP_OK) return st;
	K.up.hint_block = K.dn.hint_block = 0;
	st = e->is_complex ? build_kron_up<cplx>(e, n_up) : build_kron_up<double>(e, n_up);
	if (st != LPP_OK) return st;

	HIP_TRY_MEM(hipMalloc(&K.up_words, sizeof(uint32_t) * (size_t)n_up));
	HIP_TRY_MEM(hipMalloc(&K.dn_words, sizeof(uint32_t) * (size_t)n_dn));
	HIP_TRY_MEM(hipMalloc(&K.U, sizeof(double) * 32));
	std::vector<double> U32(32, 0.0);
	for (int i = 0; i < L; i++) U32[i] = U[i];
	HIP_TRY(hipMemcpyAsync(K.U, U32.data(), sizeof(double) * 32, hipMemcpyHostToDevice, e->stream));
	k_basis_words<<<(int)((n_up + 255) / 256), 256, 0, e->stream>>>((const uint64_t*)d_comb.p, kCombDim, n_up, nup, L, K.up_words);
	k_basis_words<<<(int)((n_dn + 255) / 256), 256, 0, e->stream>>>((const uint64_t*)d_comb.p, kCombDim, n_dn, ndown, L, K.dn_words);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (ninj) {
		// Coulomb term of HubbardOneBandExtended, 0.5 sum_ij V_ij (n_i,up + n_i,dn)(n_j,up + n_j,dn) (HubbardHelper.h:167-177), split by
		// species: a table per up word, a table per down word and the cross term sum_{i in up} (V n_dn)_i (V taken as given: i,j over all pairs)
		std::vector<uint32_t> uw((size_t)n_up), dw((size_t)n_dn);
		HIP_TRY(hipMemcpy(uw.data(), K.up_words, sizeof(uint32_t) * (size_t)n_up, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(dw.data(), K.dn_words, sizeof(uint32_t) * (size_t)n_dn, hipMemcpyDeviceToHost));
		auto same_species = [&](uint32_t w) {
			double s2 = 0.0;
			for (int i = 0; i < L; i++)
				if ((w >> i) & 1)
					for (int j = 0; j < L; j++)
						if ((w >> j) & 1) s2 += 0.5 * ninj[i * L + j];
			return s2;
		};
		std::vector<double> cu((size_t)n_up), cd((size_t)n_dn), cx((size_t)n_dn * 32, 0.0);
		for (int64_t i = 0; i < n_up; i++) cu[(size_t)i] = same_species(uw[(size_t)i]);
		for (int64_t d = 0; d < n_dn; d++) {
			cd[(size_t)d] = same_species(dw[(size_t)d]);
			for (int i = 0; i < L; i++) {
				double t = 0.0;
				for (int j = 0; j < L; j++)
					if ((dw[(size_t)d] >> j) & 1) t += 0.5 * (ninj[i * L + j] + ninj[j * L + i]);
				cx[(size_t)d * 32 + i] = t;
			}
		}
		HIP_TRY_MEM(hipMalloc(&K.cdiag_up, sizeof(double) * cu.size()));
		HIP_TRY_MEM(hipMalloc(&K.cdiag_dn, sizeof(double) * cd.size()));
		HIP_TRY_MEM(hipMalloc(&K.cross, sizeof(double) * cx.size()));
		HIP_TRY(hipMemcpy(K.cdiag_up, cu.data(), sizeof(double) * cu.size(), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(K.cdiag_dn, cd.data(), sizeof(double) * cd.size(), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(K.cross, cx.data(), sizeof(double) * cx.size(), hipMemcpyHostToDevice));
	}

	K.L = L;
	K.n_up = n_up;
	K.n_dn = n_dn;
	K.id0 = id0;
	K.nid = nid;
	K.window = (size_t)n_up * e->esz <= (size_t)156 * 1024;
	if (getenv("LPP_KRON_NO_WINDOW")) K.window = false;
	// entries of the stored CSR this product represents (one diagonal per row + all hops), for the local rows
	const double off_up = (double)K.up.nnz - (double)n_up, off_dn_total = (double)K.dn.nnz - (double)n_dn;
	K.equiv_nnz = (double)nid * ((double)n_up + off_up) + (double)n_up * off_dn_total * ((double)nid / (double)n_dn);
	if (multi && comm->exchange_begin && comm->exchange_end && comm->xchg_chunk > 0) {
		const int64_t per = (n_dn + comm->nranks - 1) / comm->nranks, peru = per > 0 ? comm->xchg_chunk / per : 0;
		if (!K.packed) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: the transposition exchange needs the packed H_up layout");
		if (!comm->send2_buf || !comm->recv2_buf || per <= 0 || comm->xchg_chunk != per * peru || peru * comm->nranks < n_up)
			return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly: transposition exchange needs send2/recv2 buffers and xchg_chunk == ceil(N_down/P) * peru, peru >= ceil(N_up/P)");
		e->tx = true;
		e->tx_per = per;
		e->tx_peru = peru;
		e->kron_n_up_tx = n_up;
	}
	K.active = true;
	e->n_local = nid * n_up;
	e->n_global = n_up * n_dn;
	e->row_start = id0 * n_up;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}


lpp_status lpp_engine_setup_hubbard_onthefly_super(lpp_engine* e, const lpp_comm* comm, int32_t L, int32_t nup, int32_t ndown,
                                                   const double* hop_re, const double* hop_im, const double* U, const double* V,
                                                   const double* ninj, const double* jcoup)
{
	bool has_j = false;
	if (jcoup && L >= 1 && L <= 31)
		for (int k = 0; k < L * L; k++) has_j |= (jcoup[k] != 0);
	if (!has_j) return lpp_engine_setup_hubbard_onthefly_ext(e, comm, L, nup, ndown, hop_re, hop_im, U, V, ninj);
	if (!e || !hop_re || !U || !V || nup < 0 || ndown < 0 || nup > L || ndown > L) return fail(LPP_ERR_INVALID, "lpp_engine_setup_hubbard_onthefly_super: bad argument (1 <= L <= 31)");
	if (comm && comm->nranks > 1) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly_super: the term-list product runs on one GPU (partition the stored matrix instead: lpp_engine_assemble_hubbard_super)");
	const std::vector<uint64_t> comb = comb_table();
	const int64_t n_up = (int64_t)binom(comb, L, nup), n_dn = (int64_t)binom(comb, L, ndown);
	bool cplx_in = false;
	if (hop_im)
		for (int k = 0; k < L * L; k++) cplx_in |= (hop_im[k] != 0);
	if (cplx_in && !e->is_complex) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly_super: complex hoppings need a c128 engine");
	if (n_up <= 0 || n_dn <= 0) return fail(LPP_ERR_INVALID, "setup_hubbard_onthefly_super: empty Hilbert space");
	HIP_TRY(hipSetDevice(e->cfg.device));
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_loc);
	free_csr(e->A_rem);
	drop_product(e);
	KronState& K = e->kron;
	std::vector<HostProc> hp;
	hubbard_terms(L, hop_re, hop_im, hp);
	super_terms(L, jcoup, hp);
	std::vector<Proc> procs;
	int nneg = 0;
	lpp_status st = finish_procs(hp, procs, &nneg);
	if (st != LPP_OK) return st;
	// the device buffers live as long as the product does (K.terms_bufs, released by free_kron)
	auto keep = [&](int slot, const void* src, size_t bytes) -> lpp_status {
		HIP_TRY_MEM(hipMalloc(&K.terms_bufs[slot], std::max<size_t>(bytes, 8)));
		if (bytes) HIP_TRY(hipMemcpyAsync(K.terms_bufs[slot], src, bytes, hipMemcpyHostToDevice, e->stream));
		return LPP_OK;
	};
	if ((st = keep(0, procs.data(), sizeof(Proc) * procs.size())) != LPP_OK) return st;
	if ((st = keep(1, comb.data(), sizeof(uint64_t) * comb.size())) != LPP_OK) return st;
	if ((st = keep(2, U, sizeof(double) * L)) != LPP_OK) return st;
	if ((st = keep(3, V, sizeof(double) * L)) != LPP_OK) return st;
	if (ninj && (st = keep(4, ninj, sizeof(double) * L * L)) != LPP_OK) return st;
	if ((st = keep(5, jcoup, sizeof(double) * L * L)) != LPP_OK) return st;
	HIP_TRY(hipStreamSynchronize(e->stream)); // the host vectors above go out of scope
	AsmParams* P = new AsmParams();
	P->model = ASM_HUBBARD;
	P->L = L;
	P->nup = nup;
	P->ndown = ndown;
	P->nproc = (int)procs.size();
	P->nneg = nneg;
	P->n_up = n_up;
	P->nrows_global = P->nloc = n_up * n_dn;
	P->row0 = 0;
	P->part = 0;
	P->procs = (const Proc*)K.terms_bufs[0];
	P->comb = (const uint64_t*)K.terms_bufs[1];
	P->d0 = (const double*)K.terms_bufs[2];
	P->d1 = (const double*)K.terms_bufs[3];
	P->d2 = (const double*)K.terms_bufs[4];
	P->d3 = (const double*)K.terms_bufs[5];
	K.terms_params = P;
	// entries of the CSR this product stands for (statistics only): one count pass
	{
		DevBuf len, sums, total;
		const int64_t n = P->nloc + 1;
		const int64_t nblk = (n + kScanChunk - 1) / kScanChunk;
		HIP_TRY_MEM(hipMalloc(&len.p, sizeof(int64_t) * (size_t)n));
		HIP_TRY(hipMemsetAsync(len.p, 0, sizeof(int64_t) * (size_t)n, e->stream));
		HIP_TRY_MEM(hipMalloc(&sums.p, sizeof(int64_t) * (size_t)nblk));
		HIP_TRY_MEM(hipMalloc(&total.p, sizeof(int64_t)));
		const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((P->nloc + kBlock - 1) / kBlock, 1 << 20));
		k_asm_count<ASM_HUBBARD><<<nb, kBlock, 0, e->stream>>>(*P, (int64_t*)len.p);
		k_scan_block_sums<<<(int)nblk, kBlock, 0, e->stream>>>((const int64_t*)len.p, n, (int64_t*)sums.p);
		k_scan_sums<<<1, kBlock, 0, e->stream>>>((int64_t*)sums.p, nblk, (int64_t*)total.p);
		int64_t nnz = 0;
		HIP_TRY(hipMemcpyAsync(&nnz, total.p, sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(e->stream));
		K.equiv_nnz = (double)nnz;
	}
	K.terms = true;
	K.L = L;
	K.n_up = n_up;
	K.n_dn = n_dn;
	K.id0 = 0;
	K.nid = n_dn;
	K.active = true;
	e->n_local = e->n_global = n_up * n_dn;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

} // extern "C"

namespace lpp {

void drop_product(lpp_engine* e)
{
	free_kron(e);
	free_tj(e);
	free_pb(e);
	e->tx = false;
}

void free_kron(lpp_engine* e)
{
	KronState& K = e->kron;
	free_csr(K.up);
	free_csr(K.dn);
	if (K.up_words) (void)hipFree(K.up_words);
	if (K.dn_words) (void)hipFree(K.dn_words);
	if (K.U) (void)hipFree(K.U);
	if (K.pk_words) (void)hipFree(K.pk_words);
	if (K.pk_off) (void)hipFree(K.pk_off);
	if (K.pk_len) (void)hipFree(K.pk_len);
	if (K.pk_dict) (void)hipFree(K.pk_dict);
	for (double* q : { K.cdiag_up, K.cdiag_dn, K.cross })
		if (q) (void)hipFree(q);
	for (void* q : K.terms_bufs)
		if (q) (void)hipFree(q);
	delete (AsmParams*)K.terms_params;
	K = KronState();
}

template <typename T> static int kron_launch_t(lpp_engine* e, const void* ywin, const void* ydown, void* x, double* partial, const EpiScale& sc, int part, int64_t b0, int64_t cnt)
{
	KronState& K = e->kron;
	if (K.terms) { // term-list product (k_asm_apply): whole vector, one GPU
		if (part != 0 || b0 != 0) return -1;
		const AsmParams& P = *(const AsmParams*)K.terms_params;
		const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((P.nloc + kBlock - 1) / kBlock, kMaxPartials));
		if (partial)
			k_asm_apply<ASM_HUBBARD, T, true><<<nb, kBlock, 0, e->stream>>>(P, (const T*)ywin, (T*)x, partial, sc);
		else
			k_asm_apply<ASM_HUBBARD, T, false><<<nb, kBlock, 0, e->stream>>>(P, (const T*)ywin, (T*)x, nullptr, sc);
		return partial ? nb : 0;
	}
	if (K.nid == 0 && part != 2) return 0;
	if (cnt < 0) cnt = K.nid - b0; // default: every block of the slice
	if (part != 2 && (b0 < 0 || b0 + cnt > K.nid)) return -1;
	if (cnt == 0 && part != 2) return 0;
	if (part != 0 && !K.packed) return -1;
	if (K.packed) {
		KronPackedArgs<T> pa;
		pa.words = K.pk_words;
		pa.slice_off = K.pk_off;
		pa.slice_len = K.pk_len;
		pa.dict = K.pk_dict;
		pa.spb = K.pk_spb;
		pa.n_up = K.n_up;
		pa.id0 = K.id0 + b0; // a sub-range of blocks: the slice pointers move with it
		pa.nid = cnt;
		pa.dn_rowptr = K.dn.rowptr;
		pa.dn_col = K.dn.col;
		pa.dn_val = (const T*)K.dn.val;
		pa.up_words = K.up_words;
		pa.dn_words = K.dn_words;
		pa.U = K.U;
		pa.cdiag_up = K.cdiag_up;
		pa.cdiag_dn = K.cdiag_dn;
		pa.cross = K.cross;
		pa.L = K.L;
		pa.ywin = ywin ? (const T*)ywin + b0 * K.n_up : nullptr;
		pa.ydown = (const T*)ydown;
		pa.x = (T*)x + (part == 2 ? 0 : b0 * K.n_up);
		pa.partial = partial;
		pa.xcd_map = (e->k2_variant >> 1) & 1;
		pa.sc = sc;
		pa.part = part;
		pa.n_dn = K.n_dn;
		bool use_window = K.window;
		if (part == 2) { // down-hops on the transposed slice: rows per down index = peru, all (padded) down indices
			pa.n_up = e->tx_peru;
			pa.id0 = 0;
			pa.nid = e->tx_per * (int64_t)e->comm.nranks;
			pa.spb = (int)((e->tx_peru + 63) / 64);
			use_window = false;
		}
		const size_t ldsb = use_window ? sizeof(T) * (size_t)std::max<int64_t>(K.n_up, 64) : 64;
		const int pcu = std::max(1, std::min(2, (int)((160 * 1024 - 8192) / (ldsb + 1))));
		int nbp = (int)std::max<int64_t>(1, std::min<int64_t>(pa.nid, (int64_t)e->num_cus * pcu));
		if (nbp >= 8) nbp &= ~7;
		const bool dotp = partial != nullptr;
		if (K.pk_nchunk > 1 && part != 2) { // N_up exceeds LDS: the source row is staged in pieces (part 2 has no up-hops)
			KronChunkArgs<T> ca;
			ca.k = pa;
			ca.nchunk = K.pk_nchunk;
			ca.cw = K.pk_cw;
			const size_t ldsc = sizeof(T) * (size_t)K.pk_cw;
			const int nbc = (int)std::max<int64_t>(1, std::min<int64_t>(pa.nid, (int64_t)e->num_cus));
			if (dotp) {
				(void)hipFuncSetAttribute((const void*)k_spmv_kron_chunked<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
				k_spmv_kron_chunked<T, true><<<nbc, kWinThreads, ldsc, e->stream>>>(ca);
			} else {
				(void)hipFuncSetAttribute((const void*)k_spmv_kron_chunked<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
				k_spmv_kron_chunked<T, false><<<nbc, kWinThreads, ldsc, e->stream>>>(ca);
			}
			return dotp ? nbc : 0;
		}
#define LPP_KP(DOT_, WIN_)                                                                                            \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_spmv_kron_packed<T, DOT_, WIN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
		k_spmv_kron_packed<T, DOT_, WIN_><<<nbp, kWinThreads, ldsb, e->stream>>>(pa);                                   \
	} while (0)
		if (dotp && use_window) LPP_KP(true, true);
		else if (dotp) LPP_KP(true, false);
		else if (use_window) LPP_KP(false, true);
		else LPP_KP(false, false);
#undef LPP_KP
		return dotp ? nbp : 0;
	}
	KronArgs<T> a;
	a.up.g = K.up.geom;
	a.up.slice_ptr = K.up.slice_ptr;
	a.up.row_len = K.up.row_len;
	a.up.col = K.up.scol;
	a.up.val = (const T*)K.up.sval;
	a.up.codes = K.up.codes;
	a.up.code_ptr = K.up.code_ptr;
	a.up.dict = K.up.dict;
	a.up.src = nullptr;
	a.up.x = nullptr;
	a.up.ydot = nullptr;
	a.up.partial = nullptr;
	a.up.xcd_map = 0;
	a.up.sc = EpiScale { nullptr, nullptr, 0 };
	a.up.dia_stride = 0;
	a.up.dia_off = nullptr;
	a.up.dia_val = nullptr;
	a.up.tmpl = 0;
	a.up.dcode = nullptr;
	a.up.tw = nullptr;
	a.up.tw_off = nullptr;
	a.up.tw_len = nullptr;
	a.up.rowmap = nullptr;
	a.n_up = K.n_up;
	a.id0 = K.id0 + b0;
	a.nid = cnt;
	a.dn_rowptr = K.dn.rowptr;
	a.dn_col = K.dn.col;
	a.dn_val = (const T*)K.dn.val;
	a.up_words = K.up_words;
	a.dn_words = K.dn_words;
	a.U = K.U;
	a.cdiag_up = K.cdiag_up;
	a.cdiag_dn = K.cdiag_dn;
	a.cross = K.cross;
	a.L = K.L;
	a.ywin = (const T*)ywin + b0 * K.n_up;
	a.ydown = (const T*)ydown;
	a.x = (T*)x + b0 * K.n_up;
	a.partial = partial;
	a.xcd_map = (e->k2_variant >> 1) & 1;
	a.sc = sc;
	const size_t lds_bytes = K.window ? sizeof(T) * (size_t)std::max<int64_t>(K.n_up, 64) : 64;
	const int per_cu = std::max(1, std::min(2, (int)((160 * 1024 - 8192) / (lds_bytes + 1))));
	int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cnt, (int64_t)e->num_cus * per_cu));
	if (nb >= 8) nb &= ~7;
	const bool dot = partial != nullptr;
	const int sel = (dot ? 4 : 0) | (K.window ? 2 : 0) | (K.up.coded ? 1 : 0);
	hipStream_t st = e->stream;
#define LPP_KRON(DOT_, WIN_, CODED_)                                                                                   \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_spmv_kron<T, DOT_, WIN_, CODED_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
		k_spmv_kron<T, DOT_, WIN_, CODED_, 8><<<nb, kWinThreads, lds_bytes, st>>>(a);                                   \
	} while (0)
	switch (sel) {
	case 0: LPP_KRON(false, false, false); break;
	case 1: LPP_KRON(false, false, true); break;
	case 2: LPP_KRON(false, true, false); break;
	case 3: LPP_KRON(false, true, true); break;
	case 4: LPP_KRON(true, false, false); break;
	case 5: LPP_KRON(true, false, true); break;
	case 6: LPP_KRON(true, true, false); break;
	default: LPP_KRON(true, true, true); break;
	}
#undef LPP_KRON
	return dot ? nb : 0;
}

int kron_launch(lpp_engine* e, const void* ywin, const void* ydown, void* x, double* partial, const EpiScale& sc, int part, int64_t b0, int64_t cnt)
{
	return e->is_complex ? kron_launch_t<cplx>(e, ywin, ydown, x, partial, sc, part, b0, cnt) : kron_launch_t<double>(e, ywin, ydown, x, partial, sc, part, b0, cnt);
}

} // namespace lpp
