// lpp_engine.hip -- the MI355X-native Lanczos inner engine behind include/lpp_engine.h.
//
// Device boundary: the CSR is made resident once, the whole
// computeAllStatesBelow loop (reference src/Engine/Engine.h:601-657 -> LanczosSolver
// [PsimagLite]) runs on the GPU, only the tridiagonal coefficients (and requested Ritz
// vectors) come back.  The host keeps the tiny tridiagonal eigenproblem and tests convergence
// `check_lag` steps behind the GPU so the stream never drains.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "lpp_assemble_kernels.h"
#include "lpp_engine_impl.h"

using namespace lpp;

namespace lpp {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
void free_csr(DevCsr& A)
{
	if (A.owned) {
		if (A.rowptr) (void)hipFree(A.rowptr);
		if (A.col) (void)hipFree(A.col);
		if (A.val) (void)hipFree(A.val);
	}
	if (A.slice_ptr) (void)hipFree(A.slice_ptr);
	if (A.row_len) (void)hipFree(A.row_len);
	if (A.scol) (void)hipFree(A.scol);
	if (A.sval) (void)hipFree(A.sval);
	if (A.codes) (void)hipFree(A.codes);
	if (A.code_ptr) (void)hipFree(A.code_ptr);
	if (A.dict) (void)hipFree(A.dict);
	if (A.rrowptr) (void)hipFree(A.rrowptr);
	if (A.dia_off) (void)hipFree(A.dia_off);
	if (A.dia_val) (void)hipFree(A.dia_val);
	if (A.dcode) (void)hipFree(A.dcode);
	if (A.tw) (void)hipFree(A.tw);
	if (A.tw_off) (void)hipFree(A.tw_off);
	if (A.tw_len) (void)hipFree(A.tw_len);
	if (A.out_part) {
		for (int q = 0; q < A.split_parts; q++) free_csr(A.out_part[q]);
		delete[] A.out_part;
	}
	if (A.out_rowmap) (void)hipFree(A.out_rowmap);
	A = DevCsr();
}

static int pick_group(int64_t nrows, int64_t nnz)
{
	if (const char* s = getenv("LPP_SPMV_G")) {
		int g = atoi(s);
		if (g == 4 || g == 8 || g == 16 || g == 32 || g == 64) return g;
	}
	const double avg = nrows > 0 ? (double)nnz / (double)nrows : 1.0;
	int g = 4;
	while (g < 64 && g * 2 <= avg / 2.0 + 1e-9) g *= 2; // largest power of two <= avg/2, in [4,64]
	return g;
}

template <typename T, bool DOT>
static void launch_rowgroup(const DevCsr& A, const T* src, T* x, const T* ydot, double* partial, int nblocks,
                            hipStream_t st, const EpiScale& sc)
{
	const int64_t* rp = A.rowptr;
	const int32_t* col = A.col;
	const T* val = (const T*)A.val;
	switch (A.G) {
	case 4: k_spmv_rowgroup<T, 4, DOT><<<nblocks, kBlock, 0, st>>>(A.nrows, rp, col, val, src, x, ydot, partial, sc); break;
	case 8: k_spmv_rowgroup<T, 8, DOT><<<nblocks, kBlock, 0, st>>>(A.nrows, rp, col, val, src, x, ydot, partial, sc); break;
	case 16: k_spmv_rowgroup<T, 16, DOT><<<nblocks, kBlock, 0, st>>>(A.nrows, rp, col, val, src, x, ydot, partial, sc); break;
	case 32: k_spmv_rowgroup<T, 32, DOT><<<nblocks, kBlock, 0, st>>>(A.nrows, rp, col, val, src, x, ydot, partial, sc); break;
	default: k_spmv_rowgroup<T, 64, DOT><<<nblocks, kBlock, 0, st>>>(A.nrows, rp, col, val, src, x, ydot, partial, sc); break;
	}
}

// x += A * src ; if partial != nullptr also partial[b] = block sums of Re<ydot|x>.
// Returns the number of partials written (0 when partial == nullptr).
template <typename T> static int spmv_launch_t(lpp_engine* e, const DevCsr& A, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc)
{
	hipStream_t st = e->stream;
	if (A.nrows == 0) return 0;
	if (A.sliced) {
		SlicedArgs<T> a;
		a.g = A.geom;
		a.slice_ptr = A.slice_ptr;
		a.row_len = A.row_len;
		a.col = A.scol;
		a.val = (const T*)A.sval;
		a.codes = A.codes;
		a.code_ptr = A.code_ptr;
		a.dict = A.dict;
		a.src = (const T*)src;
		a.x = (T*)x;
		a.ydot = (const T*)ydot;
		a.partial = partial;
		a.xcd_map = (e->k2_variant >> 1) & 1;
		a.sc = sc;
		a.dia_stride = A.dia_stride;
		a.dia_off = A.dia_off;
		a.dia_val = (const T*)A.dia_val;
		a.tmpl = A.tmpl;
		a.dcode = A.dcode;
		a.tw = A.tw;
		a.tw_off = A.tw_off;
		a.tw_len = A.tw_len;
		a.rowmap = nullptr;
		a.pad = A.pad;
		const bool dot = partial != nullptr;
		const bool u8 = (e->k2_variant & 4) != 0;
		const int sel = (dot ? 4 : 0) | (A.coded ? 2 : 0) | (u8 ? 1 : 0);
		int nb;
		if (A.window) {
			const size_t lds_bytes = sizeof(T) * (size_t)std::max<int64_t>(A.geom.B, 64);
			const int per_cu = std::max(1, std::min(2, (int)((160 * 1024 - 4096) / (lds_bytes + 1))));
			nb = (int)std::max<int64_t>(1, std::min<int64_t>(A.geom.nblocks, (int64_t)e->num_cus * per_cu));
			if (nb >= 8) nb &= ~7;
#define LPP_K3(DOT_, CODED_, U_)                                                                                      \
	do {                                                                                                              \
		if (A.local16) {                                                                                              \
			(void)hipFuncSetAttribute((const void*)k_spmv_window<T, DOT_, CODED_, U_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			k_spmv_window<T, DOT_, CODED_, U_, true><<<nb, kWinThreads, lds_bytes, st>>>(a);                            \
		} else {                                                                                                      \
			(void)hipFuncSetAttribute((const void*)k_spmv_window<T, DOT_, CODED_, U_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
			k_spmv_window<T, DOT_, CODED_, U_, false><<<nb, kWinThreads, lds_bytes, st>>>(a);                           \
		}                                                                                                             \
	} while (0)
			switch (sel) {
			case 0: LPP_K3(false, false, 4); break;
			case 1: LPP_K3(false, false, 8); break;
			case 2: LPP_K3(false, true, 4); break;
			case 3: LPP_K3(false, true, 8); break;
			case 4: LPP_K3(true, false, 4); break;
			case 5: LPP_K3(true, false, 8); break;
			case 6: LPP_K3(true, true, 4); break;
			default: LPP_K3(true, true, 8); break;
			}
#undef LPP_K3
		} else {
			const int64_t need = (A.geom.nslices + (kBlock / 64) - 1) / (kBlock / 64);
			nb = (int)std::max<int64_t>(1, std::min<int64_t>(need, e->spmv_max_blocks));
			if (nb >= 8) nb &= ~7; // multiple of 8 so the XCD-contiguous mapping applies
#define LPP_K2(DOT_, CODED_, U_) k_spmv_sliced<T, DOT_, CODED_, U_><<<nb, kBlock, 0, st>>>(a)
			switch (sel) {
			case 0: LPP_K2(false, false, 4); break;
			case 1: LPP_K2(false, false, 8); break;
			case 2: LPP_K2(false, true, 4); break;
			case 3: LPP_K2(false, true, 8); break;
			case 4: LPP_K2(true, false, 4); break;
			case 5: LPP_K2(true, false, 8); break;
			case 6: LPP_K2(true, true, 4); break;
			default: LPP_K2(true, true, 8); break;
			}
#undef LPP_K2
		}
		return partial ? nb : 0;
	}
	const int rows_per_block = kBlock / A.G;
	const int64_t need = (A.nrows + rows_per_block - 1) / rows_per_block;
	const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(need, e->spmv_max_blocks));
	if (partial)
		launch_rowgroup<T, true>(A, (const T*)src, (T*)x, (const T*)ydot, partial, nb, st, sc);
	else
		launch_rowgroup<T, false>(A, (const T*)src, (T*)x, nullptr, nullptr, nb, st, sc);
	return partial ? nb : 0;
}

// the part of a split-panel matrix that leaves the row blocks: k_spmv_sliced over the panel-major rows, one contiguous eighth
// of the panels per XCD (so every panel is gathered from ONE L2), x and ydot through the row map
template <typename T> static int spmv_launch_out_t(lpp_engine* e, const DevCsr& A, const int32_t* rowmap, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc)
{
	if (A.nrows == 0) return 0;
	SlicedArgs<T> a {};
	a.g = A.geom;
	a.slice_ptr = A.slice_ptr;
	a.row_len = A.row_len;
	a.col = A.scol;
	a.val = (const T*)A.sval;
	a.codes = A.codes;
	a.code_ptr = A.code_ptr;
	a.dict = A.dict;
	a.src = (const T*)src;
	a.x = (T*)x;
	a.ydot = (const T*)ydot;
	a.partial = partial;
	a.xcd_map = 1;
	a.sc = sc;
	a.rowmap = rowmap;
	// two resident workgroups per CU walk the panels together (measured at BASELINE config 2, scripts/experiments/r03_generic_ab.sh:
	// 512 workgroups 9.85 ms and 60.7 GB of fabric reads, 2048 11.1 ms / 66.1 GB, 4096 11.3 ms / 66.2 GB)
	const int64_t need = (A.geom.nslices + (kBlock / 64) - 1) / (kBlock / 64);
	int nb = (int)std::max<int64_t>(1, std::min<int64_t>(need, std::min<int64_t>(e->spmv_max_blocks, 2 * (int64_t)e->num_cus)));
	if (nb >= 8) nb &= ~7;
	hipStream_t st = e->stream;
	const int sel = (partial ? 2 : 0) | (A.coded ? 1 : 0);
	switch (sel) {
	case 0: k_spmv_sliced<T, false, false, 8><<<nb, kBlock, 0, st>>>(a); break;
	case 1: k_spmv_sliced<T, false, true, 8><<<nb, kBlock, 0, st>>>(a); break;
	case 2: k_spmv_sliced<T, true, false, 8><<<nb, kBlock, 0, st>>>(a); break;
	default: k_spmv_sliced<T, true, true, 8><<<nb, kBlock, 0, st>>>(a); break;
	}
	return partial ? nb : 0;
}

int spmv_launch(lpp_engine* e, const DevCsr& A, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc)
{
	if (e->tj.active && &A == &e->A_loc) return tj_launch(e, src, x, ydot, partial, sc); // t-J without a stored matrix (lpp_tj_kernels.h)
	if (A.out_part) { // split-panel layout: the in-block entries first (x = beta x + alpha A_in src), then the leaving ones on top
		if (e->is_complex) spmv_launch_t<cplx>(e, A, src, x, nullptr, nullptr, sc);
		else spmv_launch_t<double>(e, A, src, x, nullptr, nullptr, sc);
		EpiScale sc2 = sc;
		sc2.beta_one = 1;
		int np = 0;
		for (int q = 0; q < A.split_parts; q++) { // the dot rides in the last launch: x is complete there
			const bool last = q == A.split_parts - 1;
			np = e->is_complex ? spmv_launch_out_t<cplx>(e, A.out_part[q], A.out_rowmap, src, x, last ? ydot : nullptr, last ? partial : nullptr, sc2)
			                   : spmv_launch_out_t<double>(e, A.out_part[q], A.out_rowmap, src, x, last ? ydot : nullptr, last ? partial : nullptr, sc2);
		}
		return np;
	}
	return e->is_complex ? spmv_launch_t<cplx>(e, A, src, x, ydot, partial, sc) : spmv_launch_t<double>(e, A, src, x, ydot, partial, sc);
}

// distinct values of A.val -> sorted dictionary on the device; returns false when there are more than 256
template <typename T> static lpp_status try_build_dict(lpp_engine* e, DevCsr& A, const T* vals, int64_t nvals, bool* ok)
{
	*ok = false;
	if (nvals == 0) return LPP_OK;
	unsigned long long* table = nullptr;
	int* overflow = nullptr;
	HIP_TRY_MEM(hipMalloc(&table, sizeof(unsigned long long) * kDictTable));
	if (hipMalloc(&overflow, sizeof(int)) != hipSuccess) {
		(void)hipFree(table);
		return fail(LPP_ERR_NOMEM, "dictionary scratch allocation failed");
	}
	(void)hipMemsetAsync(table, 0xff, sizeof(unsigned long long) * kDictTable, e->stream);
	(void)hipMemsetAsync(overflow, 0, sizeof(int), e->stream);
	const int64_t nd = nvals * (int64_t)(sizeof(T) / sizeof(double));
	k_dict_collect<<<2048, kBlock, 0, e->stream>>>((const double*)vals, nd, table, overflow);
	std::vector<unsigned long long> host(kDictTable);
	int ov = 0;
	hipError_t e1 = hipMemcpyAsync(host.data(), table, sizeof(unsigned long long) * kDictTable, hipMemcpyDeviceToHost, e->stream);
	hipError_t e2 = hipMemcpyAsync(&ov, overflow, sizeof(int), hipMemcpyDeviceToHost, e->stream);
	hipError_t e3 = hipStreamSynchronize(e->stream);
	(void)hipFree(table);
	(void)hipFree(overflow);
	if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(LPP_ERR_HIP, "dictionary collection failed");
	std::vector<unsigned long long> keys;
	keys.push_back(0ull); // code 0 is reserved for +0.0: padded / inactive slots decode to an exact zero
	for (unsigned long long k : host)
		if (k != kDictEmpty && k != 0ull) keys.push_back(k);
	if (ov || keys.size() > 256) return LPP_OK;
	std::sort(keys.begin(), keys.end());
	std::vector<double> dict(256);
	for (size_t i = 0; i < 256; i++) {
		const unsigned long long k = keys[std::min(i, keys.size() - 1)];
		std::memcpy(&dict[i], &k, sizeof(double));
	}
	HIP_TRY_MEM(hipMalloc(&A.dict, sizeof(double) * 256));
	HIP_TRY(hipMemcpyAsync(A.dict, dict.data(), sizeof(double) * 256, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	A.ndict = (int)keys.size();
	*ok = true;
	return LPP_OK;
}

// LPP_VERBOSE=1: wall-clock of the one-off layout stages on stderr
struct StageTimer {
	const char* what;
	bool on;
	std::chrono::steady_clock::time_point t0;
	explicit StageTimer(const char* w) : what(w), on(getenv("LPP_VERBOSE") != nullptr), t0(std::chrono::steady_clock::now()) { }
	~StageTimer()
	{
		if (on) fprintf(stderr, "lpp: %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	}
};

// in-place exclusive scan of arr[0..n) (arr[n-1] must be 0 on entry and receives the total, also returned)
static lpp_status scan_exclusive(lpp_engine* e, int64_t* arr, int64_t n, int64_t* total_out)
{
	const int64_t nblk = (n + kScanChunk - 1) / kScanChunk;
	int64_t *sums = nullptr, *total = nullptr;
	HIP_TRY_MEM(hipMalloc(&sums, sizeof(int64_t) * (size_t)nblk));
	if (hipMalloc(&total, sizeof(int64_t)) != hipSuccess) {
		(void)hipFree(sums);
		return fail(LPP_ERR_NOMEM, "scan scratch allocation failed");
	}
	k_scan_block_sums<<<(int)nblk, kBlock, 0, e->stream>>>(arr, n, sums);
	k_scan_sums<<<1, kBlock, 0, e->stream>>>(sums, nblk, total);
	k_scan_apply<<<(int)nblk, kBlock, 0, e->stream>>>(arr, n, sums, arr);
	hipError_t e1 = hipMemcpyAsync(total_out, total, sizeof(int64_t), hipMemcpyDeviceToHost, e->stream);
	hipError_t e2 = hipStreamSynchronize(e->stream);
	(void)hipFree(sums);
	(void)hipFree(total);
	if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "device scan failed");
	return LPP_OK;
}

static void drop_dia(DevCsr& A)
{
	for (void* p : { (void*)A.rrowptr, (void*)A.dia_off, A.dia_val, (void*)A.dcode })
		if (p) (void)hipFree(p);
	A.rrowptr = nullptr;
	A.dcode = nullptr;
	A.dia_off = nullptr;
	A.dia_val = nullptr;
	A.rnnz = A.ndia = 0;
	A.dia_stride = 0;
}

// Split the shared-offset entries off the plain CSR of A (see k_dia_split).  On success with A.rrowptr != nullptr the
// rest CSR is (A.rrowptr, *rcol, *rval) -- the caller frees rcol / rval -- and A.dia_* hold the shared lists.
// Leaves A untouched (rrowptr == nullptr) when rows are unsorted or fewer than 10 % of the entries are shared.
// xdiag: also split the diagonal off (needs the value dictionary: its codes go to A.dcode, one per real component and row)
template <typename T> static lpp_status split_dia_t(lpp_engine* e, DevCsr& A, const SliceGeom& g, bool for_window, bool xdiag, int32_t** rcol, T** rval)
{
	*rcol = nullptr;
	*rval = nullptr;
	unsigned long long* flags = nullptr; // [0] unsorted, [1] max shared per slice, [2] total shared, [3] rows without a diagonal
	HIP_TRY_MEM(hipMalloc(&flags, sizeof(unsigned long long) * 4));
	struct Free {
		void* p;
		~Free() { (void)hipFree(p); }
	} free_flags { flags };
	HIP_TRY(hipMemsetAsync(flags, 0, sizeof(unsigned long long) * 4, e->stream));
	unsigned long long h[4] = { 0, 0, 0, 0 };
	if (!A.known_sorted) {
		k_rows_sorted<<<(int)((A.nrows + 255) / 256), 256, 0, e->stream>>>(A.nrows, A.rowptr, A.col, (int*)flags);
		HIP_TRY(hipMemcpyAsync(h, flags, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		if (h[0] != 0) return LPP_OK;
	}
	HIP_TRY_MEM(hipMalloc(&A.rrowptr, sizeof(int64_t) * (size_t)(A.nrows + 1)));
	HIP_TRY(hipMemsetAsync(A.rrowptr, 0, sizeof(int64_t) * (size_t)(A.nrows + 1), e->stream));
	const int nbw = (int)std::max<int64_t>(1, std::min<int64_t>((g.nslices + 3) / 4, 16384));
	const int win = for_window ? 1 : 0;
	int xd = xdiag ? 1 : 0;
	StageTimer* t_count = new StageTimer("  split: count pass");
	for (;;) {
		k_dia_split<T, false><<<nbw, kBlock, 0, e->stream>>>(g, A.rowptr, A.col, (const T*)A.val, win, 0, A.rrowptr, flags + 1, nullptr, nullptr,
		                                                    nullptr, nullptr, nullptr, xd, nullptr, 0, nullptr);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(h, flags, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		if (!xd || h[3] == 0) break;
		// some row has no diagonal entry: count again with the diagonal left where it is
		xd = 0;
		HIP_TRY(hipMemsetAsync(A.rrowptr, 0, sizeof(int64_t) * (size_t)(A.nrows + 1), e->stream));
		HIP_TRY(hipMemsetAsync(flags + 1, 0, sizeof(unsigned long long) * 3, e->stream));
	}
	delete t_count;
	lpp_status st = scan_exclusive(e, A.rrowptr, A.nrows + 1, &A.rnnz);
	if (st != LPP_OK) return st;
	A.ndia = (int64_t)h[2];
	A.dia_stride = (int)((h[1] + 3) & ~3ull);
	if (getenv("LPP_VERBOSE"))
		fprintf(stderr, "lpp: shared-offset split: nnz %lld -> per-row %lld + %lld per-slice entries (%lld slices, <= %d per slice)%s\n",
		        (long long)A.nnz, (long long)A.rnnz, (long long)A.ndia, (long long)g.nslices, (int)h[1], xd ? " + diagonal codes" : "");
	if ((double)(A.nnz - A.rnnz) < 0.10 * (double)A.nnz || (A.dia_stride == 0 && !xd)) {
		drop_dia(A);
		return LPP_OK;
	}
	if (xd) {
		HIP_TRY_MEM(hipMalloc(&A.dcode, (size_t)A.nrows * (sizeof(T) / sizeof(double))));
		HIP_TRY(hipMemsetAsync(A.dcode, 0, (size_t)A.nrows * (sizeof(T) / sizeof(double)), e->stream));
	}
	StageTimer* t_alloc = new StageTimer("  split: allocations");
	const size_t places = std::max<size_t>((size_t)g.nslices * (size_t)A.dia_stride, 1);
	HIP_TRY_MEM(hipMalloc(rcol, sizeof(int32_t) * (size_t)std::max<int64_t>(A.rnnz, 1)));
	HIP_TRY_MEM(hipMalloc(rval, sizeof(T) * (size_t)std::max<int64_t>(A.rnnz, 1)));
	HIP_TRY_MEM(hipMalloc(&A.dia_off, sizeof(int32_t) * places));
	HIP_TRY_MEM(hipMalloc(&A.dia_val, sizeof(T) * places));
	HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)A.dia_off, (int)kDiaNone, places, e->stream));
	HIP_TRY(hipMemsetAsync(A.dia_val, 0, sizeof(T) * places, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	delete t_alloc;
	StageTimer t_fill("  split: fill pass");
	k_dia_split<T, true><<<nbw, kBlock, 0, e->stream>>>(g, A.rowptr, A.col, (const T*)A.val, win, A.dia_stride, nullptr, nullptr, A.rrowptr,
	                                                   *rcol, *rval, A.dia_off, (T*)A.dia_val, xd, A.dict, A.ndict, A.dcode);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	return LPP_OK;
}

// build the sliced layout of A on the device for row blocks of B rows
template <typename T> static lpp_status build_sliced_t(lpp_engine* e, DevCsr& A, int64_t B, bool for_window)
{
	SliceGeom g;
	g.nrows = A.nrows;
	g.B = std::max<int64_t>(1, std::min<int64_t>(B, std::max<int64_t>(A.nrows, 1)));
	g.spb = (int32_t)((g.B + 63) / 64);
	g.nblocks = (A.nrows + g.B - 1) / g.B;
	g.nslices = g.nblocks * g.spb;
	A.geom = g;
	// entries shared by all rows of a slice are split off first; the sliced arrays then hold the rest
	const int64_t* rp = A.rowptr;
	const int32_t* cc = A.col;
	const T* vv = (const T*)A.val;
	int64_t nz = A.nnz;
	int32_t* rcol = nullptr;
	T* rval = nullptr;
	struct Scratch {
		int32_t*& c;
		T*& v;
		~Scratch()
		{
			if (c) (void)hipFree(c);
			if (v) (void)hipFree(v);
		}
	} scratch { rcol, rval };
	// the value dictionary is built from the whole matrix first: the diagonal codes below need it
	int want = e->cfg.compress_values;
	if (const char* s = getenv("LPP_COMPRESS_VALUES")) want = atoi(s);
	bool coded = false;
	if (want != 0) {
		StageTimer tm("value dictionary");
		lpp_status st = try_build_dict<T>(e, A, vv, nz, &coded);
		if (st != LPP_OK) return st;
	}
	int want_dia = A.no_dia ? 0 : 1;
	if (const char* s = getenv("LPP_SHARED_OFFSETS")) want_dia = A.no_dia ? 0 : atoi(s);
	const bool xdiag = for_window && coded && !(getenv("LPP_DIAG_CODES") && atoi(getenv("LPP_DIAG_CODES")) == 0);
	if (want_dia && A.nnz > 0) {
		StageTimer tm("shared-offset split");
		lpp_status st = split_dia_t<T>(e, A, g, for_window, xdiag, &rcol, &rval);
		if (st != LPP_OK) return st;
		if (A.rrowptr) {
			rp = A.rrowptr;
			cc = rcol;
			vv = rval;
			nz = A.rnnz;
		}
	}
	// window kernel: 16-bit window-local columns when every per-row entry stays inside its row block
	bool l16 = false;
	if (for_window && g.B <= 65536 && nz > 0 && !(getenv("LPP_LOCAL16") && atoi(getenv("LPP_LOCAL16")) == 0)) {
		int* outside = nullptr;
		HIP_TRY_MEM(hipMalloc(&outside, sizeof(int)));
		(void)hipMemsetAsync(outside, 0, sizeof(int), e->stream);
		k_cols_local<<<(int)((A.nrows + 255) / 256), 256, 0, e->stream>>>(g, rp, cc, outside);
		int bad = 0;
		hipError_t e1 = hipMemcpyAsync(&bad, outside, sizeof(int), hipMemcpyDeviceToHost, e->stream);
		hipError_t e2 = hipStreamSynchronize(e->stream);
		(void)hipFree(outside);
		if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "column locality check failed");
		l16 = !bad;
	}
	A.local16 = l16;
	// +64 entries of slack: the pipelined kernel reads (and discards) the entry after a slice's last one
	HIP_TRY_MEM(hipMalloc(&A.slice_ptr, sizeof(int64_t) * (size_t)(g.nslices + 1)));
	HIP_TRY_MEM(hipMalloc(&A.row_len, sizeof(int32_t) * (size_t)std::max<int64_t>(A.nrows, 1)));
	const size_t colsz = l16 ? sizeof(uint16_t) : sizeof(int32_t);
	HIP_TRY_MEM(hipMalloc(&A.scol, colsz * (size_t)(nz + 64)));
	HIP_TRY(hipMemsetAsync((char*)A.scol + colsz * (size_t)nz, 0, colsz * 64, e->stream));
	if (coded) HIP_TRY_MEM(hipMalloc(&A.code_ptr, sizeof(int64_t) * (size_t)(g.nslices + 1)));
	const int64_t nthreads = std::max<int64_t>(A.nrows, g.nslices + 1);
	const int nb = (int)((nthreads + 255) / 256);
	k_slice_meta<<<nb, 256, 0, e->stream>>>(g, rp, A.slice_ptr, A.row_len, A.code_ptr, CodeTraits<T>::kSlotsPerWord);
	const int64_t need = (g.nslices + 3) / 4;
	const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>(need, 8192));
	if (coded) {
		// exclusive scan of the per-slice word counts -> code_ptr; the grand total sizes the code array
		int64_t nwords = 0;
		lpp_status st = scan_exclusive(e, A.code_ptr, g.nslices + 1, &nwords);
		if (st != LPP_OK) return st;
		A.code_words = nwords;
		HIP_TRY_MEM(hipMalloc(&A.codes, sizeof(uint32_t) * (size_t)(nwords + 64 * 16)));
		HIP_TRY(hipMemsetAsync(A.codes + nwords, 0, sizeof(uint32_t) * 64 * 16, e->stream));
		if (l16)
			k_slice_fill<T, false, true><<<nb2, kBlock, 0, e->stream>>>(g, rp, cc, (const T*)nullptr, A.scol, (T*)nullptr);
		else
			k_slice_fill<T, false><<<nb2, kBlock, 0, e->stream>>>(g, rp, cc, (const T*)nullptr, A.scol, (T*)nullptr);
		k_slice_codes<T><<<nb2, kBlock, 0, e->stream>>>(g, rp, vv, A.code_ptr, A.dict, A.ndict, A.codes);
		A.coded = true;
	} else {
		HIP_TRY_MEM(hipMalloc(&A.sval, sizeof(T) * (size_t)(nz + 64)));
		HIP_TRY(hipMemsetAsync((T*)A.sval + nz, 0, sizeof(T) * 64, e->stream));
		// plain values, window kernel, no shared-offset split: the entries that leave a row's block go into the row's first slots, so that
		// every 512-byte run of the source vector is gathered by one load (k_slice_fill); needs rows sorted by column
		bool outs = for_window && !l16 && rp == A.rowptr && g.nblocks >= 2 && nz > 0 && !(getenv("LPP_OUTS_FIRST") && atoi(getenv("LPP_OUTS_FIRST")) == 0);
		if (outs && !A.known_sorted) {
			int* unsorted = nullptr;
			HIP_TRY_MEM(hipMalloc(&unsorted, sizeof(int)));
			(void)hipMemsetAsync(unsorted, 0, sizeof(int), e->stream);
			k_rows_sorted<<<(int)((A.nrows + 255) / 256), 256, 0, e->stream>>>(A.nrows, A.rowptr, A.col, unsorted);
			int bad = 1;
			hipError_t e1 = hipMemcpyAsync(&bad, unsorted, sizeof(int), hipMemcpyDeviceToHost, e->stream);
			hipError_t e2 = hipStreamSynchronize(e->stream);
			(void)hipFree(unsorted);
			if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "row order check failed");
			outs = bad == 0;
		}
		A.outs_first = outs;
		// ... and the vectors pitched to 128-byte lines per row block (whole matrix on one GPU, real, blocks not line-aligned by themselves, from
		// 256 MB per vector on): the runs gathered from other blocks then start on a line -- 4 lines per run instead of 4.9 at config 2
		// (N_up = 12870 = 6 mod 16).  LPP_PITCH_ROWS=0/1: never / wherever it applies.
		int pad = 0;
		{
			const int64_t line = 128 / (int64_t)sizeof(T);
			bool want = (size_t)A.nrows * sizeof(T) >= ((size_t)256 << 20);
			if (const char* sp = getenv("LPP_PITCH_ROWS")) want = atoi(sp) != 0;
			const int64_t pitched = (g.B + line - 1) / line * line;
			if (outs && want && &A == &e->A_loc && !A.out_part && !e->has_comm && !e->is_complex && A.src_elems == 0 && g.nrows == g.nblocks * g.B && g.B % line != 0
			    && pitched * g.nblocks < ((int64_t)1 << 31) && (size_t)(pitched * g.nblocks) * sizeof(T) < ((size_t)1 << 32))
				pad = (int)(pitched - g.B);
		}
		A.pad = pad;
		if (pad) {
			e->pitch = g.B + pad;
			e->pitch_rows = g.B;
			e->pitch_blocks = g.nblocks;
		}
		if (l16)
			k_slice_fill<T, false, true><<<nb2, kBlock, 0, e->stream>>>(g, rp, cc, vv, A.scol, (T*)A.sval);
		else
			k_slice_fill<T, false><<<nb2, kBlock, 0, e->stream>>>(g, rp, cc, vv, A.scol, (T*)A.sval, 0, outs ? 1 : 0, pad);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	A.sliced = true;
	// Block-periodic structure: when every row block repeats block 0's row lengths and local columns (the in-block
	// part of a product basis is the same one-species matrix in every block) only block 0's copy is kept; it stays
	// in L2 instead of streaming 2 bytes per entry + 4 per row from HBM.  Values (codes) remain per block.
	if (l16 && coded && g.nblocks >= 2 && g.nrows == g.nblocks * g.B && !(getenv("LPP_BLOCK_TEMPLATE") && atoi(getenv("LPP_BLOCK_TEMPLATE")) == 0)) {
		int* differs = nullptr;
		HIP_TRY_MEM(hipMalloc(&differs, sizeof(int) * 2));
		(void)hipMemsetAsync(differs, 0, sizeof(int) * 2, e->stream);
		k_tmpl_check<<<nb2, kBlock, 0, e->stream>>>(g, A.slice_ptr, A.row_len, (const uint16_t*)A.scol, A.code_ptr, A.codes, differs);
		int hd[2] = { 1, 1 };
		hipError_t e1 = hipMemcpyAsync(hd, differs, sizeof(int) * 2, hipMemcpyDeviceToHost, e->stream);
		hipError_t e2 = hipStreamSynchronize(e->stream);
		(void)hipFree(differs);
		if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "block-template check failed");
		const int bad = hd[0];
		if (!bad && !hd[1]) {
			// the value codes repeat as well (the diagonal travels apart): keep block 0's code words only
			int64_t w0 = 0;
			HIP_TRY(hipMemcpy(&w0, A.code_ptr + g.spb, sizeof(int64_t), hipMemcpyDeviceToHost));
			int64_t* cp = nullptr;
			uint32_t* cw = nullptr;
			HIP_TRY_MEM(hipMalloc(&cp, sizeof(int64_t) * (size_t)(g.spb + 1)));
			HIP_TRY_MEM(hipMalloc(&cw, sizeof(uint32_t) * (size_t)(w0 + 64 * 16)));
			HIP_TRY(hipMemcpyAsync(cp, A.code_ptr, sizeof(int64_t) * (size_t)(g.spb + 1), hipMemcpyDeviceToDevice, e->stream));
			HIP_TRY(hipMemcpyAsync(cw, A.codes, sizeof(uint32_t) * (size_t)w0, hipMemcpyDeviceToDevice, e->stream));
			HIP_TRY(hipMemsetAsync(cw + w0, 0, sizeof(uint32_t) * 64 * 16, e->stream));
			HIP_TRY(hipStreamSynchronize(e->stream));
			(void)hipFree(A.code_ptr);
			(void)hipFree(A.codes);
			A.code_ptr = cp;
			A.codes = cw;
			A.code_words = w0;
		}
		if (!bad) {
			int64_t n0 = 0; // entries of block 0
			HIP_TRY(hipMemcpy(&n0, A.slice_ptr + g.spb, sizeof(int64_t), hipMemcpyDeviceToHost));
			int64_t* sp = nullptr;
			int32_t* rl = nullptr;
			int32_t* sc = nullptr;
			HIP_TRY_MEM(hipMalloc(&sp, sizeof(int64_t) * (size_t)(g.spb + 1)));
			HIP_TRY_MEM(hipMalloc(&rl, sizeof(int32_t) * (size_t)g.B));
			HIP_TRY_MEM(hipMalloc(&sc, sizeof(uint16_t) * (size_t)(n0 + 64)));
			HIP_TRY(hipMemcpyAsync(sp, A.slice_ptr, sizeof(int64_t) * (size_t)(g.spb + 1), hipMemcpyDeviceToDevice, e->stream));
			HIP_TRY(hipMemcpyAsync(rl, A.row_len, sizeof(int32_t) * (size_t)g.B, hipMemcpyDeviceToDevice, e->stream));
			HIP_TRY(hipMemcpyAsync(sc, A.scol, sizeof(uint16_t) * (size_t)n0, hipMemcpyDeviceToDevice, e->stream));
			HIP_TRY(hipMemsetAsync((char*)sc + sizeof(uint16_t) * (size_t)n0, 0, sizeof(uint16_t) * 64, e->stream));
			HIP_TRY(hipStreamSynchronize(e->stream));
			(void)hipFree(A.slice_ptr);
			(void)hipFree(A.row_len);
			(void)hipFree(A.scol);
			A.slice_ptr = sp;
			A.row_len = rl;
			A.scol = sc;
			A.tmpl = hd[1] ? 1 : 2;
			if (A.tmpl == 2 && !(getenv("LPP_TEMPLATE_PACK") && atoi(getenv("LPP_TEMPLATE_PACK")) == 0)) {
				// packed padded copy of the template for the inner loop (the compact copy stays for get_csr)
				HIP_TRY_MEM(hipMalloc(&A.tw_len, sizeof(int32_t) * (size_t)g.spb));
				HIP_TRY_MEM(hipMalloc(&A.tw_off, sizeof(int32_t) * (size_t)g.spb));
				const int nbp = (int)std::max<int64_t>(1, (g.spb + 3) / 4);
				k_tmpl_pack<T, 0><<<nbp, kBlock, 0, e->stream>>>(g, A.slice_ptr, A.row_len, (const uint16_t*)A.scol, A.code_ptr, A.codes, A.tw_len,
				                                                 nullptr, nullptr);
				std::vector<int32_t> hl((size_t)g.spb), ho((size_t)g.spb);
				HIP_TRY(hipMemcpyAsync(hl.data(), A.tw_len, sizeof(int32_t) * (size_t)g.spb, hipMemcpyDeviceToHost, e->stream));
				HIP_TRY(hipStreamSynchronize(e->stream));
				int64_t tot = 0;
				for (int32_t j2 = 0; j2 < g.spb; j2++) {
					ho[(size_t)j2] = (int32_t)tot;
					tot += (int64_t)hl[(size_t)j2] * 64;
				}
				if (tot < ((int64_t)1 << 30)) {
					HIP_TRY_MEM(hipMalloc(&A.tw, sizeof(uint32_t) * (size_t)std::max<int64_t>(tot, 1)));
					HIP_TRY(hipMemcpyAsync(A.tw_off, ho.data(), sizeof(int32_t) * (size_t)g.spb, hipMemcpyHostToDevice, e->stream));
					k_tmpl_pack<T, 1><<<nbp, kBlock, 0, e->stream>>>(g, A.slice_ptr, A.row_len, (const uint16_t*)A.scol, A.code_ptr, A.codes, A.tw_len,
					                                                 A.tw_off, A.tw);
					HIP_TRY(hipGetLastError());
					HIP_TRY(hipStreamSynchronize(e->stream));
				}
			}
		}
	}
	return LPP_OK;
}

// Basis block of an uploaded matrix nobody described (lpp_engine_set_row_block not called): product bases show up as
// couplings at offsets k*B that are identical for 64 consecutive rows.  A sample of 64-row slices is copied to the
// host; per slice the largest offset shared by all its rows (same value in every row) is a block shift, B = the gcd of
// those over the sample.  Accepted only when every sampled entry is then either inside its row's block or a whole
// number of blocks away, and B fits the LDS window.  Returns 0 when no such structure is found.
template <typename T> static int64_t detect_row_block_t(lpp_engine* e, const DevCsr& A, int64_t lds_cap_elems)
{
	const int64_t nsl = A.nrows / 64;
	if (nsl < 64 || !A.col || !A.val || A.nnz == 0) return 0;
	const int samples = 48;
	std::vector<int64_t> rp(65);
	std::vector<int32_t> ci;
	std::vector<T> va;
	struct Slice {
		int64_t row0;
		std::vector<int64_t> rp;
		std::vector<int32_t> ci;
	};
	std::vector<Slice> kept;
	int64_t G = 0;
	for (int sidx = 0; sidx < samples; sidx++) {
		const int64_t row0 = ((nsl * (2 * sidx + 1)) / (2 * samples)) * 64;
		if (hipMemcpy(rp.data(), A.rowptr + row0, sizeof(int64_t) * 65, hipMemcpyDeviceToHost) != hipSuccess) return 0;
		const int64_t p0 = rp[0], n = rp[64] - rp[0];
		if (n <= 0 || n > (int64_t)1 << 20) continue;
		ci.resize((size_t)n);
		va.resize((size_t)n);
		if (hipMemcpy(ci.data(), A.col + p0, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
		if (hipMemcpy(va.data(), (const T*)A.val + p0, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
		int64_t best = 0; // largest |offset| shared by all 64 rows with the same value
		for (int64_t q = rp[0]; q < rp[1]; q++) {
			const int64_t off = (int64_t)ci[(size_t)(q - p0)] - row0;
			if (std::llabs(off) < 64 || std::llabs(off) <= best) continue;
			bool all = true;
			for (int r = 1; r < 64 && all; r++) {
				bool found = false;
				for (int64_t t = rp[r]; t < rp[r + 1]; t++)
					if ((int64_t)ci[(size_t)(t - p0)] - (row0 + r) == off) {
						found = std::memcmp(&va[(size_t)(t - p0)], &va[(size_t)(q - p0)], sizeof(T)) == 0;
						break;
					}
				all = found;
			}
			if (all) best = std::llabs(off);
		}
		if (best > 0) G = G == 0 ? best : std::gcd(G, best);
		Slice sl;
		sl.row0 = row0;
		sl.rp.assign(rp.begin(), rp.end());
		sl.ci = ci;
		kept.push_back(std::move(sl));
	}
	if (G < 64 || G > lds_cap_elems || kept.empty()) return 0;
	for (const Slice& sl : kept) {
		const int64_t p0 = sl.rp[0];
		for (int r = 0; r < 64; r++) {
			const int64_t row = sl.row0 + r, b0 = (row / G) * G;
			for (int64_t t = sl.rp[r]; t < sl.rp[r + 1]; t++) {
				const int64_t c = sl.ci[(size_t)(t - p0)];
				if (!((c >= b0 && c < b0 + G) || (c - row) % G == 0)) return 0;
			}
		}
	}
	return G;
}

// Split-panel layout (k_split_count): A keeps the entries inside its row blocks, *A.out_part takes the others in panel-major row
// order.  Only for plain-format matrices (shared offsets off: they take the leaving entries out of the rows in their own way),
// rows sorted by column, whole blocks, and a leaving part worth a second launch (>= 20 % of the entries).
template <typename T> static lpp_status split_panel_t(lpp_engine* e, DevCsr& A, int64_t B, bool* did)
{
	*did = false;
	const int64_t nb = A.nrows / B;
	if (nb < 64 || nb * B != A.nrows || !A.col || !A.val || A.nnz == 0) return LPP_OK;
	hipStream_t st = e->stream;
	if (!A.known_sorted) {
		int* flag = nullptr;
		HIP_TRY_MEM(hipMalloc(&flag, sizeof(int)));
		(void)hipMemsetAsync(flag, 0, sizeof(int), st);
		k_rows_sorted<<<(int)((A.nrows + 255) / 256), 256, 0, st>>>(A.nrows, A.rowptr, A.col, flag);
		int h = 1;
		hipError_t e1 = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, st);
		hipError_t e2 = hipStreamSynchronize(st);
		(void)hipFree(flag);
		if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "split-panel layout: sortedness check failed");
		if (h) return LPP_OK;
	}
	StageTimer tm("split-panel layout");
	// parts by source block range: what one part gathers from while a panel is walked -- two 128-byte lines per source block
	// when the rows are not line-aligned (B not a multiple of 16), one when they are -- is kept near 1.7 MB
	SplitParams P;
	P.nrows = A.nrows;
	P.B = B;
	P.nb = nb;
	const size_t foot = (size_t)nb * ((B & 15) ? 256 : 128);
	P.nparts = (int)std::max<size_t>(1, std::min<size_t>(kSplitMaxParts, (foot + ((size_t)1700 << 10) - 1) / ((size_t)1700 << 10)));
	P.nparts = 1; // measured: parts do not raise the L2 hit rate here (2 parts: 6.0 + 6.2 ms and 32 + 34 GB against 9.9 ms and 61 GB in one)
	(void)foot;
	if (const char* s2 = getenv("LPP_SPLIT_PARTS")) P.nparts = std::max(1, std::min(atoi(s2), kSplitMaxParts));
	P.pblk = (nb + P.nparts - 1) / P.nparts;
	const int nbk = (int)((A.nrows + 255) / 256);
	DevCsr* O = new DevCsr[P.nparts];
	struct Guard {
		DevCsr*& o;
		int n;
		int64_t* rpi = nullptr;
		int32_t* ci = nullptr;
		void* vi = nullptr;
		int32_t* map = nullptr;
		void* ptrs = nullptr;
		~Guard()
		{
			if (o) {
				for (int q = 0; q < n; q++) free_csr(o[q]);
				delete[] o;
			}
			for (void* p : { (void*)rpi, (void*)ci, vi, (void*)map, ptrs })
				if (p) (void)hipFree(p);
		}
	} g { O, P.nparts };
	HIP_TRY_MEM(hipMalloc(&g.rpi, sizeof(int64_t) * (size_t)(A.nrows + 1)));
	HIP_TRY_MEM(hipMalloc(&g.map, sizeof(int32_t) * (size_t)A.nrows));
	HIP_TRY(hipMemsetAsync(g.rpi, 0, sizeof(int64_t) * (size_t)(A.nrows + 1), st));
	int64_t* lens[kSplitMaxParts] = { nullptr, nullptr, nullptr, nullptr };
	for (int q = 0; q < P.nparts; q++) {
		O[q].nrows = A.nrows;
		O[q].owned = true;
		O[q].known_sorted = true;
		O[q].no_dia = true;
		HIP_TRY_MEM(hipMalloc(&O[q].rowptr, sizeof(int64_t) * (size_t)(A.nrows + 1)));
		HIP_TRY(hipMemsetAsync(O[q].rowptr, 0, sizeof(int64_t) * (size_t)(A.nrows + 1), st));
		lens[q] = O[q].rowptr;
	}
	HIP_TRY_MEM(hipMalloc(&g.ptrs, sizeof(lens)));
	HIP_TRY(hipMemcpyAsync(g.ptrs, lens, sizeof(lens), hipMemcpyHostToDevice, st));
	k_split_count<<<nbk, 256, 0, st>>>(P, A.rowptr, A.col, g.rpi, (int64_t* const*)g.ptrs, g.map);
	int64_t nin = 0, nout = 0;
	lpp_status rc = scan_exclusive(e, g.rpi, A.nrows + 1, &nin);
	if (rc != LPP_OK) return rc;
	for (int q = 0; q < P.nparts; q++) {
		int64_t n = 0;
		rc = scan_exclusive(e, O[q].rowptr, A.nrows + 1, &n);
		if (rc != LPP_OK) return rc;
		O[q].nnz = n;
		nout += n;
	}
	if (nin + nout != A.nnz) return fail(LPP_ERR_HIP, "split-panel layout: entry count mismatch");
	if ((double)nout < 0.20 * (double)A.nnz) return LPP_OK; // not worth the extra launches
	HIP_TRY_MEM(hipMalloc(&g.ci, sizeof(int32_t) * (size_t)std::max<int64_t>(nin, 1)));
	HIP_TRY_MEM(hipMalloc(&g.vi, sizeof(T) * (size_t)std::max<int64_t>(nin, 1)));
	SplitOut SO {};
	for (int q = 0; q < P.nparts; q++) {
		HIP_TRY_MEM(hipMalloc(&O[q].col, sizeof(int32_t) * (size_t)std::max<int64_t>(O[q].nnz, 1)));
		HIP_TRY_MEM(hipMalloc(&O[q].val, sizeof(T) * (size_t)std::max<int64_t>(O[q].nnz, 1)));
		SO.rp[q] = O[q].rowptr;
		SO.col[q] = O[q].col;
		SO.val[q] = O[q].val;
	}
	k_split_fill<T><<<nbk, 256, 0, st>>>(P, A.rowptr, A.col, (const T*)A.val, g.rpi, g.ci, (T*)g.vi, SO);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	// the in-block part replaces the matrix in A
	(void)hipFree(A.rowptr);
	(void)hipFree(A.col);
	(void)hipFree(A.val);
	A.rowptr = g.rpi;
	A.col = g.ci;
	A.val = g.vi;
	g.rpi = nullptr;
	g.ci = nullptr;
	g.vi = nullptr;
	A.nnz = nin;
	A.known_sorted = true;
	// the leaving parts: sliced layout over ALL rows in panel-major order (one "block" = the whole matrix)
	for (int q = 0; q < P.nparts; q++) {
		rc = finalize_csr(e, O[q], true, LPP_SPMV_SLICED, 0);
		if (rc != LPP_OK) return rc;
	}
	A.out_part = O;
	A.split_parts = P.nparts;
	g.o = nullptr;
	O = nullptr;
	A.out_rowmap = g.map;
	g.map = nullptr;
	A.split_B = B;
	A.split_nb = nb;
	A.split_pblk = P.pblk;
	*did = true;
	return LPP_OK;
}

// Choose the SpMV kernel and build its layout.  hint_block (rows) is the basis' natural block
// (N_up for the Hubbard product basis): when a whole number of such blocks fits the LDS window,
// the window kernel serves every in-block gather (diagonal + up-hops) from LDS.
lpp_status finalize_csr(lpp_engine* e, DevCsr& A, bool allow_drop_plain, int force_mode, int64_t force_block)
{
	A.G = pick_group(A.nrows, A.nnz);
	int mode = e->cfg.spmv_kernel;
	if (const char* s = getenv("LPP_SPMV_KERNEL")) mode = atoi(s);
	if (force_mode) mode = force_mode;
	// the sliced kernels address the source vector with 32-bit byte offsets
	const size_t src_elems = (size_t)std::max<int64_t>(A.src_elems, A.nrows);
	const bool fits32 = src_elems * e->esz < ((size_t)1 << 32);
	const int64_t lds_cap_elems = (int64_t)((156 * 1024) / e->esz);
	int64_t win_rows = 0;
	if (A.hint_block == 0 && mode == LPP_SPMV_AUTO && !e->is_complex && fits32 && A.src_elems == 0
	    && !(getenv("LPP_DETECT_BLOCK") && atoi(getenv("LPP_DETECT_BLOCK")) == 0)) {
		StageTimer tm("basis block detection");
		A.hint_block = detect_row_block_t<double>(e, A, lds_cap_elems);
		if (A.hint_block && getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: detected a basis block of %lld rows\n", (long long)A.hint_block);
	}
	if (A.hint_block > 0 && A.hint_block <= lds_cap_elems) {
		// one basis block per window whenever it gives the 16 waves of a workgroup something to do: only then do all
		// row blocks repeat the same in-block structure (block template); tiny basis blocks are grouped
		const int64_t k = A.hint_block >= 512 ? 1 : std::max<int64_t>(1, std::min<int64_t>(lds_cap_elems / A.hint_block, (1024 + A.hint_block - 1) / A.hint_block));
		win_rows = A.hint_block * k;
	}
	if (mode == LPP_SPMV_AUTO) {
		// the LDS window pays when in-block gathers dominate (Hubbard up-hops), the matrix is large enough to give every
		// CU several blocks, and a window element is 8 bytes (measured: complex t-J windows lose).  With the block
		// template and 16-bit local columns it also wins when the whole vector fits the Infinity Cache
		// (Hubbard chains L=12: 31.6 vs 33.1 us, L=14: 0.306 vs 0.372 ms).
		bool window_ok = win_rows > 0 && !e->is_complex && (A.nrows + win_rows - 1) / win_rows >= 2 * (int64_t)e->num_cus;
		if (!window_ok && win_rows == 0 && !e->is_complex && fits32 && A.col && A.nnz > 0) {
			// no natural block: a plain diagonal window of 16384 rows pays when enough gathers land inside it (bases in
			// ascending word order couple mostly nearby ranks: Heisenberg L=28 0.95 vs 1.22 ms); measured here
			const int64_t gw = std::min<int64_t>(lds_cap_elems, 16384);
			if ((A.nrows + gw - 1) / gw >= 2 * (int64_t)e->num_cus) {
				unsigned long long* inside = nullptr;
				HIP_TRY_MEM(hipMalloc(&inside, sizeof(unsigned long long)));
				(void)hipMemsetAsync(inside, 0, sizeof(unsigned long long), e->stream);
				k_count_local<<<(int)((A.nrows + 255) / 256), 256, 0, e->stream>>>(A.nrows, gw, A.rowptr, A.col, inside);
				unsigned long long h = 0;
				hipError_t e1 = hipMemcpyAsync(&h, inside, sizeof(h), hipMemcpyDeviceToHost, e->stream);
				hipError_t e2 = hipStreamSynchronize(e->stream);
				(void)hipFree(inside);
				if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "window locality count failed");
				if ((double)h >= 0.35 * (double)A.nnz) {
					window_ok = true;
					win_rows = gw;
				}
			}
		}
		mode = window_ok ? LPP_SPMV_WINDOW : LPP_SPMV_SLICED;
	}
	if (mode == LPP_SPMV_WINDOW && win_rows == 0) {
		// no natural block: a generic diagonal window (captures near-diagonal columns)
		win_rows = std::min<int64_t>(lds_cap_elems, 16384);
	}
	if (!force_mode)
		if (const char* s = getenv("LPP_WINDOW_ROWS")) win_rows = std::max<int64_t>(64, std::min<int64_t>(atoll(s), lds_cap_elems));
	if (!fits32 && (mode == LPP_SPMV_AUTO || mode == LPP_SPMV_SLICED || mode == LPP_SPMV_WINDOW)) mode = LPP_SPMV_ROWGROUP;
	if ((mode == LPP_SPMV_SLICED || mode == LPP_SPMV_WINDOW) && A.nrows > 0) {
		const int64_t B = force_block > 0 ? force_block : ((mode == LPP_SPMV_WINDOW) ? win_rows : A.nrows);
		const bool win = (mode == LPP_SPMV_WINDOW);
		// plain-format matrix with a basis block: the entries that leave the row blocks go panel-major (split_panel_t)
		{
			int want_dia = A.no_dia ? 0 : 1;
			if (const char* s2 = getenv("LPP_SHARED_OFFSETS")) want_dia = A.no_dia ? 0 : atoi(s2);
			// Taken by itself where a panel of 16 positions is ONE 128-byte line of every source block -- basis blocks of a multiple of 16
			// rows (vectors are not pitched in the general layout) -- from 256 MB per vector on.  Measured on the 4x4 cluster with
			// 7 + 7 electrons (N_up = 11440 = 16 x 715, 1.3e8 rows, 58.6 GB algorithmic): 6.2 + 7.0 = 13.2 ms and 72 GB of fabric reads in two
			// launches against 13.9 ms / 85.6 GB in one kernel (4.45 against 4.2 TB/s algorithmic = 0.56 of 8 TB/s; both launches stream at
			// 5.2-5.75 TB/s: the 12-byte format's ceiling on this matrix).  At BASELINE config 2 (N_up = 12870 = 6 mod 16: two lines per
			// source block and panel, which no longer fit the L2 beside the entry streams) it loses, 8.1 + 9.9 ms against 18.3-18.7 ms in
			// one kernel, and is not taken.  LPP_SPLIT_PANEL=0 / 1: never / wherever a basis block is known.
			// (real matrices only: 16 complex positions are two lines, and no complex shape has been measured)
			bool off = e->is_complex || (A.hint_block & 15) != 0 || (size_t)A.nrows * e->esz < ((size_t)256 << 20);
			if (const char* sp = getenv("LPP_SPLIT_PANEL")) off = atoi(sp) == 0;
			if (win && !force_mode && !want_dia && !off && A.hint_block > 0 && B == A.hint_block && !A.out_part && A.src_elems == 0) {
				bool did = false;
				lpp_status st2 = e->is_complex ? split_panel_t<cplx>(e, A, B, &did) : split_panel_t<double>(e, A, B, &did);
				if (st2 != LPP_OK) return st2;
			}
		}
		StageTimer tm("sliced layout (total)");
		lpp_status st = e->is_complex ? build_sliced_t<cplx>(e, A, B, win) : build_sliced_t<double>(e, A, B, win);
		if (st != LPP_OK) return st;
		A.window = (mode == LPP_SPMV_WINDOW);
		if (allow_drop_plain && getenv("LPP_KEEP_PLAIN_CSR") == nullptr && A.owned) {
			StageTimer tm("release plain CSR");
			// the sliced copy is the resident one; release the plain arrays (frees ~12 B/nnz)
			(void)hipFree(A.col);
			(void)hipFree(A.val);
			A.col = nullptr;
			A.val = nullptr;
		}
	}
	return LPP_OK;
}

void set_spmv_bytes(lpp_engine* e)
{
	const double s = (double)e->esz;
	const double N = (double)e->n_local;
	if (e->pb.active) {
		// the CSR this layout stands for (several GPUs: this rank's share of the rows)
		e->spmv_bytes = (double)e->pb.nnz_loc * (s + 4.0) + (N + 1.0) * 8.0 + 3.0 * N * s;
		return;
	}
	if (e->tj.active) { // the CSR the hole-major t-J form stands for
		e->spmv_bytes = (double)e->tj.nnz * (s + 4.0) + (N + 1.0) * 8.0 + 3.0 * N * s;
		return;
	}
	if (e->kron.active) {
		// matrix-free product: no matrix stream.  Vector-streaming model: x in/out and y once (3 N s) plus one
		// coalesced pass over the source block of every connected down-configuration (H_down off-diagonals).
		const KronState& K = e->kron;
		if (K.terms) { // term-list product: the CSR it stands for, as the SURVEY 8(d) figure
			e->spmv_bytes = K.equiv_nnz * (s + 4.0) + (N + 1.0) * 8.0 + 3.0 * N * s;
			return;
		}
		const double avg_down = K.n_dn > 0 ? ((double)K.dn.nnz - (double)K.n_dn) / (double)K.n_dn : 0.0;
		e->spmv_bytes = N * s * (3.0 + avg_down);
		return;
	}
	const double Z = (double)(e->A_loc.nnz + e->A_loc.out_nnz() + e->A_rem.nnz);
	e->spmv_bytes = Z * (s + 4.0) + (N + 1.0) * 8.0 + 3.0 * N * s;
}

lpp_status alloc_work(lpp_engine* e)
{
	// vectors are padded to an even number of doubles so BLAS-1 kernels can move double2
	// product-basis matrices keep their vectors pitched (block b at element b*pitch, padding zero)
	const int64_t nd = (e->pitch > 0 ? e->pitch * e->pitch_blocks : e->n_local) * (e->is_complex ? 2 : 1);
	e->nd = nd;
	e->nd_pad = (nd + 1) & ~(int64_t)1;
	e->n2 = e->nd_pad / 2;
	for (double** p : { &e->x, &e->y }) {
		if (*p) (void)hipFree(*p);
		*p = nullptr;
		HIP_TRY_MEM(hipMalloc(p, sizeof(double) * (size_t)std::max<int64_t>(e->nd_pad, 2)));
		HIP_TRY(hipMemsetAsync(*p, 0, sizeof(double) * (size_t)std::max<int64_t>(e->nd_pad, 2), e->stream));
	}
	return LPP_OK;
}

} // namespace lpp

// device scratch released on scope exit
struct DevScratch {
	void* p = nullptr;
	DevScratch() = default;
	DevScratch(const DevScratch&) = delete;
	DevScratch& operator=(const DevScratch&) = delete;
	~DevScratch()
	{
		if (p) (void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)); }
	void take(DevScratch& o)
	{
		if (p) (void)hipFree(p);
		p = o.p;
		o.p = nullptr;
	}
};

// Plain CSR order (columns, values) of a matrix whose only resident form is the sliced layout: undo the slot-major
// order (template-aware), decode the value codes, then merge the per-slice shared entries and the diagonal codes back.
template <typename T> static lpp_status rebuild_csr_t(lpp_engine* e, const DevCsr& A, DevScratch& col_out, DevScratch& val_out)
{
	const int64_t* rp = A.rrowptr ? A.rrowptr : A.rowptr; // the sliced arrays hold the rest CSR when entries were split off
	const int64_t nz = A.rrowptr ? A.rnnz : A.nnz;
	DevScratch tcol, tval;
	if (tcol.alloc(sizeof(int32_t) * (size_t)nz) != hipSuccess || tval.alloc(sizeof(T) * (size_t)nz) != hipSuccess)
		return fail(LPP_ERR_NOMEM, "lpp_engine_get_csr: scratch allocation failed");
	const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>((A.geom.nslices + 3) / 4, 8192));
	T* plain_vals = A.coded ? nullptr : (T*)tval.p;
	if (A.local16)
		k_slice_fill<T, true, true><<<nb2, kBlock, 0, e->stream>>>(A.geom, rp, A.scol, (const T*)A.sval, (int32_t*)tcol.p, plain_vals, A.tmpl ? 1 : 0);
	else
		k_slice_fill<T, true><<<nb2, kBlock, 0, e->stream>>>(A.geom, rp, A.scol, (const T*)A.sval, (int32_t*)tcol.p, plain_vals, 0, A.outs_first ? 1 : 0, A.pad);
	if (A.coded) k_slice_decode<T><<<nb2, kBlock, 0, e->stream>>>(A.geom, rp, A.codes, A.code_ptr, A.dict, (T*)tval.p, A.tmpl == 2 ? 1 : 0);
	if (A.rrowptr) {
		DevScratch fcol, fval;
		if (fcol.alloc(sizeof(int32_t) * (size_t)A.nnz) != hipSuccess || fval.alloc(sizeof(T) * (size_t)A.nnz) != hipSuccess)
			return fail(LPP_ERR_NOMEM, "lpp_engine_get_csr: scratch allocation failed");
		k_dia_merge<T><<<(int)((A.nrows + 255) / 256), 256, 0, e->stream>>>(A.geom, A.rowptr, rp, (const int32_t*)tcol.p, (const T*)tval.p, A.dia_stride,
		                                                                   A.dia_off, (const T*)A.dia_val, (int32_t*)fcol.p, (T*)fval.p, A.dcode, A.dict);
		HIP_TRY(hipStreamSynchronize(e->stream)); // the merge reads tcol / tval, which are released next
		tcol.take(fcol);
		tval.take(fval);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	col_out.take(tcol);
	val_out.take(tval);
	return LPP_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

lpp_status lpp_engine_create(lpp_engine** out, const lpp_config* cfg)
{
	if (!out || !cfg) return fail(LPP_ERR_INVALID, "lpp_engine_create: null argument");
	if (cfg->abi_version != LPP_ABI_VERSION) return fail(LPP_ERR_INVALID, "lpp_engine_create: ABI version mismatch");
	if (cfg->dtype != LPP_F64 && cfg->dtype != LPP_C128) return fail(LPP_ERR_INVALID, "lpp_engine_create: bad dtype");
	if (cfg->max_steps < 1) return fail(LPP_ERR_INVALID, "lpp_engine_create: max_steps < 1");
	int ndev = 0;
	hipError_t err = hipGetDeviceCount(&ndev);
	if (err != hipSuccess || ndev <= 0)
		return fail(LPP_ERR_HIP, std::string("lpp_engine_create: no HIP device (there is no CPU fallback): ") + hipGetErrorString(err));
	if (cfg->device < 0 || cfg->device >= ndev) return fail(LPP_ERR_INVALID, "lpp_engine_create: device ordinal out of range");
	HIP_TRY(hipSetDevice(cfg->device));
	lpp_engine* e = new lpp_engine();
	e->cfg = *cfg;
	if (e->cfg.check_lag < 0) e->cfg.check_lag = 0;
	if (e->cfg.check_lag > 16) e->cfg.check_lag = 16;
	e->is_complex = (cfg->dtype == LPP_C128);
	e->esz = e->is_complex ? 16 : 8;
	if (cfg->stream) {
		e->stream = (hipStream_t)cfg->stream;
		e->own_stream = false;
	} else {
		err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
		if (err != hipSuccess) {
			delete e;
			return fail(LPP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(err));
		}
		e->own_stream = true;
	}
	e->spmv_max_blocks = 256 * 16;
	// bit1: XCD-contiguous slice/block map, bit2: 8 slots per batch.  Measured (profiles/README.md): the plain
	// round-robin map is faster for every real-valued workload (4x4 Hubbard 10.8 vs 11.6 ms, matrix-free 4.47 vs
	// 4.70, Heisenberg L=28 1.29 vs 1.36), the XCD-contiguous map for the complex t-J matrix (0.62 vs 0.68).
	e->k2_variant = e->is_complex ? 6 : 4;
	if (const char* s = getenv("LPP_K2_VARIANT")) e->k2_variant = atoi(s);
	{
		int ncu = 0;
		if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) e->num_cus = ncu;
	}
	if (const char* s = getenv("LPP_SPMV_BLOCKS")) e->spmv_max_blocks = std::max(1, std::min(atoi(s), kMaxPartials));
	const int M = cfg->max_steps + 2;
	e->M = M;
	err = hipMalloc(&e->partial, sizeof(double) * (size_t)kMaxPartials * 2 * kPanel);
	if (err == hipSuccess) err = hipMalloc(&e->scal_own, sizeof(double) * (size_t)(6 * M + 8));
	if (err == hipSuccess) err = hipHostMalloc(&e->h_scal, sizeof(double) * (size_t)(2 * M + 2), hipHostMallocDefault);
	if (err != hipSuccess) {
		lpp_engine_destroy(e);
		return fail(LPP_ERR_NOMEM, std::string("lpp_engine_create: allocation failed: ") + hipGetErrorString(err));
	}
	e->bind_scalars(e->scal_own);
	(void)hipEventCreate(&e->ev_t0);
	(void)hipEventCreate(&e->ev_t1);
	*out = e;
	return LPP_OK;
}

void* lpp_engine_stream(lpp_engine* e) { return e ? (void*)e->stream : nullptr; }

lpp_status lpp_engine_destroy(lpp_engine* e)
{
	if (!e) return LPP_OK;
	(void)hipSetDevice(e->cfg.device);
	if (e->stream) (void)hipStreamSynchronize(e->stream);
	free_csr(e->A_loc);
	free_csr(e->A_rem);
	drop_product(e);
	free_pb(e);
	for (double* p : { e->x, e->y, e->V, e->partial, e->scal_own, e->zwork })
		if (p) (void)hipFree(p);
	if (e->h_scal) (void)hipHostFree(e->h_scal);
	for (auto& ev : e->step_events)
		if (ev) (void)hipEventDestroy(ev);
	for (auto& pr : e->spmv_events) {
		(void)hipEventDestroy(pr.first);
		(void)hipEventDestroy(pr.second);
	}
	if (e->ev_t0) (void)hipEventDestroy(e->ev_t0);
	if (e->ev_t1) (void)hipEventDestroy(e->ev_t1);
	if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
	delete e;
	return LPP_OK;
}

// A whole matrix (real, or complex Hermitian with a real diagonal) on one GPU that turns out to be of product-basis form (basis block = the caller's hint or the detected one)
// is taken into the product-basis layout (pb_from_csr: T, C, D extracted and the CSR verified against them row by row) and the
// CSR is dropped; *as_product says so.  Everything else goes on to finalize_csr.
static lpp_status try_product_layout(lpp_engine* e, DevCsr& A, bool* as_product)
{
	*as_product = false;
	if (A.nrows == 0 || A.nnz == 0) return LPP_OK;
	if (e->hint.kind) { // the caller described the model behind this matrix (lpp_engine_set_model_*): its structured form, if the description is this CSR
		StageTimer tm("structured form from the model description");
		bool done = false;
		lpp_status st = model_layout_from_hint(e, A, &done);
		e->hint = ModelHint(); // one matrix per description
		if (st != LPP_OK) return st;
		if (done) {
			*as_product = true;
			free_csr(A);
			return LPP_OK;
		}
	}
	int64_t hb = A.hint_block;
	if (hb == 0 && !(getenv("LPP_DETECT_BLOCK") && atoi(getenv("LPP_DETECT_BLOCK")) == 0) && (size_t)A.nrows * e->esz < ((size_t)1 << 32)) {
		StageTimer tm("basis block detection");
		hb = e->is_complex ? detect_row_block_t<cplx>(e, A, (int64_t)1 << 23) : detect_row_block_t<double>(e, A, (int64_t)1 << 23);
		if (hb && getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: detected a basis block of %lld rows\n", (long long)hb);
		if (hb > 0 && hb <= (int64_t)((156 * 1024) / e->esz)) A.hint_block = hb; // finalize_csr need not look again
	}
	if (hb <= 0) return LPP_OK;
	StageTimer tm("product-basis layout from the CSR");
	lpp_status st = pb_from_csr(e, A, hb, as_product);
	if (st != LPP_OK) return st;
	if (*as_product) free_csr(A);
	return LPP_OK;
}

static lpp_status upload_csr(lpp_engine* e, DevCsr& A, int64_t nrows, const int64_t* rowptr, const int32_t* colind, const void* values,
                             int64_t hint_block = 0, bool try_product = false)
{
	const int64_t keep_src = A.src_elems;
	free_csr(A);
	A.src_elems = keep_src;
	A.hint_block = hint_block;
	A.nrows = nrows;
	A.nnz = rowptr ? rowptr[nrows] : 0;
	A.owned = true;
	HIP_TRY_MEM(hipMalloc(&A.rowptr, sizeof(int64_t) * (size_t)(nrows + 1)));
	HIP_TRY_MEM(hipMalloc(&A.col, sizeof(int32_t) * (size_t)std::max<int64_t>(A.nnz, 1)));
	HIP_TRY_MEM(hipMalloc(&A.val, e->esz * (size_t)std::max<int64_t>(A.nnz, 1)));
	HIP_TRY(hipMemcpyAsync(A.rowptr, rowptr, sizeof(int64_t) * (size_t)(nrows + 1), hipMemcpyHostToDevice, e->stream));
	if (A.nnz > 0) {
		HIP_TRY(hipMemcpyAsync(A.col, colind, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(A.val, values, e->esz * (size_t)A.nnz, hipMemcpyHostToDevice, e->stream));
	}
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (try_product) {
		bool as_product = false;
		lpp_status st = try_product_layout(e, A, &as_product);
		if (st != LPP_OK || as_product) return st;
	}
	return finalize_csr(e, A, true);
}

static lpp_status check_csr_host(int64_t nrows, int64_t ncols, const int64_t* rowptr, const int32_t* colind)
{
	if (rowptr[0] != 0) return fail(LPP_ERR_INVALID, "CSR: rowptr[0] != 0");
	for (int64_t i = 0; i < nrows; i++)
		if (rowptr[i + 1] < rowptr[i]) return fail(LPP_ERR_INVALID, "CSR: rowptr not monotone");
	const int64_t nnz = rowptr[nrows];
	for (int64_t p = 0; p < nnz; p++)
		if (colind[p] < 0 || (int64_t)colind[p] >= ncols) return fail(LPP_ERR_INVALID, "CSR: column index out of range");
	return LPP_OK;
}

lpp_status lpp_engine_set_row_block(lpp_engine* e, int64_t rows_per_block)
{
	if (!e || rows_per_block < 0) return fail(LPP_ERR_INVALID, "lpp_engine_set_row_block: bad argument");
	e->row_block_hint = rows_per_block;
	return LPP_OK;
}

lpp_status lpp_engine_set_model_tj(lpp_engine* e, int32_t L, int32_t nup, int32_t ndown, const double* hop_re, const double* hop_im, const double* jpm,
                                   const double* jzz, const double* w, const double* potentialV, int32_t npot)
{
	if (!e) return fail(LPP_ERR_INVALID, "lpp_engine_set_model_tj: null engine");
	e->hint = ModelHint();
	if (L == 0) return LPP_OK; // forget the description
	if (!hop_re || !jpm || !jzz || !w || L < 1 || L > 31 || nup < 0 || ndown < 0 || nup + ndown > L || (potentialV && npot > 0 && npot < 2 * L))
		return fail(LPP_ERR_INVALID, "lpp_engine_set_model_tj: bad argument");
	TjModel& M = e->hint.tj;
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
	e->hint.kind = 1;
	return LPP_OK;
}

lpp_status lpp_engine_set_model_heisenberg(lpp_engine* e, int32_t L, int32_t szPlusConst, const double* jpm, const double* jzz, const double* field, int32_t nfield)
{
	if (!e) return fail(LPP_ERR_INVALID, "lpp_engine_set_model_heisenberg: null engine");
	e->hint = ModelHint();
	if (L == 0) return LPP_OK;
	if (!jpm || !jzz || L < 1 || L > 62 || szPlusConst < 0 || szPlusConst > L || (nfield > 0 && !field)) return fail(LPP_ERR_INVALID, "lpp_engine_set_model_heisenberg: bad argument");
	ModelHint& H = e->hint;
	H.L = L;
	H.m = szPlusConst;
	H.nfield = std::max(nfield, 0);
	H.jpm.assign(jpm, jpm + (size_t)L * L);
	H.jzz.assign(jzz, jzz + (size_t)L * L);
	if (nfield > 0) H.field.assign(field, field + nfield);
	H.kind = 2;
	return LPP_OK;
}

lpp_status lpp_engine_set_csr(lpp_engine* e, int64_t nrows, const int64_t* rowptr, const int32_t* colind, const void* values)
{
	if (!e || nrows < 0 || !rowptr) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr: bad argument");
	if (nrows > (int64_t)INT32_MAX) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr: nrows exceeds 32-bit column range");
	if (rowptr[nrows] > 0 && (!colind || !values)) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr: null colind/values");
	lpp_status st = check_csr_host(nrows, nrows, rowptr, colind);
	if (st != LPP_OK) return st;
	HIP_TRY(hipSetDevice(e->cfg.device));
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_rem);
	drop_product(e);
	st = upload_csr(e, e->A_loc, nrows, rowptr, colind, values, e->row_block_hint, true);
	if (st != LPP_OK) return st;
	e->n_local = e->n_global = nrows;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

lpp_status lpp_engine_set_csr_device(lpp_engine* e, int64_t nrows, const int64_t* d_rowptr, const int32_t* d_colind, const void* d_values)
{
	if (!e || nrows < 0 || !d_rowptr) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr_device: bad argument");
	if (nrows > (int64_t)INT32_MAX) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr_device: nrows exceeds 32-bit column range");
	HIP_TRY(hipSetDevice(e->cfg.device));
	int64_t nnz = 0;
	HIP_TRY(hipMemcpy(&nnz, d_rowptr + nrows, sizeof(int64_t), hipMemcpyDeviceToHost));
	if (nnz < 0) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr_device: negative rowptr[nrows]");
	if (nnz > 0 && (!d_colind || !d_values)) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr_device: null colind/values");
	// the same structural checks as the host entry point, on the device
	int* bad = nullptr;
	HIP_TRY_MEM(hipMalloc(&bad, sizeof(int)));
	(void)hipMemsetAsync(bad, 0, sizeof(int), e->stream);
	k_check_csr<<<(int)((std::max<int64_t>(nrows, 1) + 255) / 256), 256, 0, e->stream>>>(nrows, nrows, d_rowptr, d_colind, bad);
	int hbad = 0;
	hipError_t e1 = hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, e->stream);
	hipError_t e2 = hipStreamSynchronize(e->stream);
	(void)hipFree(bad);
	if (e1 != hipSuccess || e2 != hipSuccess) return fail(LPP_ERR_HIP, "lpp_engine_set_csr_device: validation failed to run");
	if (hbad) return fail(LPP_ERR_INVALID, "CSR: rowptr not monotone from 0 or column index out of range");
	e->has_comm = false;
	e->bind_scalars(e->scal_own);
	free_csr(e->A_rem);
	drop_product(e);
	DevCsr& A = e->A_loc;
	free_csr(A);
	A.hint_block = e->row_block_hint;
	A.nrows = nrows;
	A.nnz = nnz;
	A.owned = true;
	HIP_TRY_MEM(hipMalloc(&A.rowptr, sizeof(int64_t) * (size_t)(nrows + 1)));
	HIP_TRY_MEM(hipMalloc(&A.col, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
	HIP_TRY_MEM(hipMalloc(&A.val, e->esz * (size_t)std::max<int64_t>(nnz, 1)));
	HIP_TRY(hipMemcpyAsync(A.rowptr, d_rowptr, sizeof(int64_t) * (size_t)(nrows + 1), hipMemcpyDeviceToDevice, e->stream));
	if (nnz > 0) {
		HIP_TRY(hipMemcpyAsync(A.col, d_colind, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(A.val, d_values, e->esz * (size_t)nnz, hipMemcpyDeviceToDevice, e->stream));
	}
	HIP_TRY(hipStreamSynchronize(e->stream));
	bool as_product = false;
	lpp_status st = try_product_layout(e, A, &as_product);
	if (st != LPP_OK) return st;
	if (!as_product) st = finalize_csr(e, A, true);
	if (st != LPP_OK) return st;
	e->n_local = e->n_global = nrows;
	e->row_start = 0;
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

lpp_status lpp_engine_set_csr_partition(lpp_engine* e, const lpp_comm* comm, int64_t global_rows, const int64_t* shard_starts,
                                        const int64_t* rowptr, const int32_t* colind, const void* values)
{
	if (!e || !comm || !shard_starts || !rowptr) return fail(LPP_ERR_INVALID, "lpp_engine_set_csr_partition: bad argument");
	lpp_status st = e->adopt_comm(comm);
	if (st != LPP_OK) return st;
	const int32_t r = comm->rank, P = comm->nranks;
	if (shard_starts[0] != 0 || shard_starts[P] != global_rows) return fail(LPP_ERR_INVALID, "set_csr_partition: shard_starts must span [0, global_rows]");
	const int64_t local = shard_starts[r + 1] - shard_starts[r];
	st = check_csr_host(local, global_rows, rowptr, colind);
	if (st != LPP_OK) return st;
	HIP_TRY(hipSetDevice(e->cfg.device));
	int64_t nl = 0, nr = 0;
	st = lpp_split_csr(r, P, shard_starts, comm->shard_stride, local, rowptr, colind, values, (int32_t)e->esz, &nl, &nr, nullptr,
	                   nullptr, nullptr, nullptr, nullptr, nullptr);
	if (st != LPP_OK) return st;
	std::vector<int64_t> rpl(local + 1), rpr(local + 1);
	std::vector<int32_t> cl(std::max<int64_t>(nl, 1)), cr(std::max<int64_t>(nr, 1));
	std::vector<char> vl((size_t)std::max<int64_t>(nl, 1) * e->esz), vr((size_t)std::max<int64_t>(nr, 1) * e->esz);
	st = lpp_split_csr(r, P, shard_starts, comm->shard_stride, local, rowptr, colind, values, (int32_t)e->esz, &nl, &nr, rpl.data(),
	                   cl.data(), vl.data(), rpr.data(), cr.data(), vr.data());
	if (st != LPP_OK) return st;
	drop_product(e);
	// the hint survives the partition when this rank's rows start on a block boundary (columns of A_loc are local)
	const int64_t hint = (e->row_block_hint > 0 && shard_starts[r] % e->row_block_hint == 0) ? e->row_block_hint : 0;
	st = upload_csr(e, e->A_loc, local, rpl.data(), cl.data(), vl.data(), hint);
	if (st != LPP_OK) return st;
	e->A_rem.src_elems = (int64_t)comm->nranks * comm->shard_stride;
	st = upload_csr(e, e->A_rem, local, rpr.data(), cr.data(), vr.data());
	if (st != LPP_OK) return st;
	e->n_local = local;
	e->n_global = global_rows;
	e->row_start = shard_starts[r];
	e->active = false;
	set_spmv_bytes(e);
	return alloc_work(e);
}

lpp_status lpp_engine_get_csr(lpp_engine* e, int32_t which, int64_t* nrows, int64_t* nnz, int64_t* rowptr, int32_t* colind, void* values)
{
	if (!e) return fail(LPP_ERR_INVALID, "lpp_engine_get_csr: null engine");
	DevCsr& A = which == 0 ? e->A_loc : e->A_rem;
	if (e->tj.active || (e->pb.active && e->pb.chain_model)) {
		// t-J without a stored matrix / a spin chain planned from its couplings: the device assembler runs again, in the reference's order
		const int64_t have = e->tj.active ? e->tj.nnz : e->pb.nnz;
		if (nrows) *nrows = which == 0 ? e->n_local : 0;
		if (nnz) *nnz = which == 0 ? have : 0;
		if (!rowptr && !colind && !values) return LPP_OK;
		if (which != 0) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: no remote part");
		HIP_TRY(hipSetDevice(e->cfg.device));
		DevCsr T;
		const PbState& B = e->pb;
		lpp_status st = e->tj.active ? assemble_tj_raw(e, e->tj.model, T)
		                             : assemble_heisenberg_raw(e, B.chain_L, B.chain_m, B.chain_jpm.data(), B.chain_jzz.data(),
		                                                       B.chain_field.empty() ? nullptr : B.chain_field.data(), B.chain_nfield, T);
		if (st == LPP_OK && T.nnz != have) st = fail(LPP_ERR_HIP, "lpp_engine_get_csr: the regenerated matrix has a different number of entries");
		hipError_t he = hipSuccess;
		if (st == LPP_OK && rowptr) he = hipMemcpy(rowptr, T.rowptr, sizeof(int64_t) * (size_t)(T.nrows + 1), hipMemcpyDeviceToHost);
		if (st == LPP_OK && he == hipSuccess && colind) he = hipMemcpy(colind, T.col, sizeof(int32_t) * (size_t)T.nnz, hipMemcpyDeviceToHost);
		if (st == LPP_OK && he == hipSuccess && values) he = hipMemcpy(values, T.val, e->esz * (size_t)T.nnz, hipMemcpyDeviceToHost);
		free_csr(T);
		if (st != LPP_OK) return st;
		if (he != hipSuccess) return fail(LPP_ERR_HIP, std::string("lpp_engine_get_csr: ") + hipGetErrorString(he));
		return LPP_OK;
	}
	if (e->pb.active) { // product-basis layout: the CSR is regenerated from T, C and the diagonal codes
		if (e->pb.tx) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: not available for the product-basis layout on several GPUs");
		if (nrows) *nrows = which == 0 ? e->n_local : 0;
		if (nnz) *nnz = which == 0 ? e->pb.nnz : 0;
		if (!rowptr && !colind && !values) return LPP_OK;
		if (which != 0) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: no remote part");
		HIP_TRY(hipSetDevice(e->cfg.device));
		return pb_get_csr(e, rowptr, colind, values);
	}
	const int64_t nnz_all = A.nnz + A.out_nnz();
	if (nrows) *nrows = A.nrows;
	if (nnz) *nnz = nnz_all;
	if (!rowptr && !colind && !values) return LPP_OK;
	if (e->kron.active) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: the matrix-free engine stores no CSR");
	if (!A.rowptr) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: no matrix");
	HIP_TRY(hipSetDevice(e->cfg.device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	// plain (col, val) arrays of one stored CSR on the device: the resident ones, or rebuilt from the sliced layout
	auto plain_of = [&](const DevCsr& X, DevScratch& scol, DevScratch& sval, const int32_t*& dcol, const void*& dval) -> lpp_status {
		dcol = X.col;
		dval = X.val;
		if (X.nnz == 0 || (dcol && dval)) return LPP_OK;
		if (!X.sliced) return fail(LPP_ERR_STATE, "lpp_engine_get_csr: matrix arrays missing");
		lpp_status st = e->is_complex ? rebuild_csr_t<cplx>(e, X, scol, sval) : rebuild_csr_t<double>(e, X, scol, sval);
		if (st != LPP_OK) return st;
		dcol = (const int32_t*)scol.p;
		dval = sval.p;
		return LPP_OK;
	};
	if (A.out_part) {
		// split-panel layout: the in-block CSR and the panel-major CSRs of the leaving entries, merged back row by row
		SplitParams P;
		P.nrows = A.nrows;
		P.B = A.split_B;
		P.nb = A.split_nb;
		P.nparts = A.split_parts;
		P.pblk = A.split_pblk;
		DevScratch ic, iv, oc[kSplitMaxParts], ov[kSplitMaxParts], mrp, mc, mv;
		const int32_t* dic = nullptr;
		const void* div = nullptr;
		lpp_status st = plain_of(A, ic, iv, dic, div);
		if (st != LPP_OK) return st;
		SplitOut SO {};
		for (int q = 0; q < A.split_parts; q++) {
			const int32_t* dc = nullptr;
			const void* dv = nullptr;
			if ((st = plain_of(A.out_part[q], oc[q], ov[q], dc, dv)) != LPP_OK) return st;
			SO.rp[q] = A.out_part[q].rowptr;
			SO.col[q] = (int32_t*)dc;
			SO.val[q] = (void*)dv;
		}
		if (mrp.alloc(sizeof(int64_t) * (size_t)(A.nrows + 1)) != hipSuccess || mc.alloc(sizeof(int32_t) * (size_t)nnz_all) != hipSuccess
		    || mv.alloc(e->esz * (size_t)nnz_all) != hipSuccess)
			return fail(LPP_ERR_NOMEM, "lpp_engine_get_csr: scratch allocation failed");
		HIP_TRY(hipMemsetAsync(mrp.p, 0, sizeof(int64_t) * (size_t)(A.nrows + 1), e->stream));
		const int nbk = (int)((A.nrows + 255) / 256);
		k_split_lengths<<<nbk, 256, 0, e->stream>>>(P, A.rowptr, SO, (int64_t*)mrp.p);
		int64_t tot = 0;
		if ((st = scan_exclusive(e, (int64_t*)mrp.p, A.nrows + 1, &tot)) != LPP_OK) return st;
		if (tot != nnz_all) return fail(LPP_ERR_HIP, "lpp_engine_get_csr: split-panel merge lost entries");
		if (e->is_complex)
			k_split_merge<cplx><<<nbk, 256, 0, e->stream>>>(P, A.rowptr, dic, (const cplx*)div, SO, (const int64_t*)mrp.p, (int32_t*)mc.p, (cplx*)mv.p);
		else
			k_split_merge<double><<<nbk, 256, 0, e->stream>>>(P, A.rowptr, dic, (const double*)div, SO, (const int64_t*)mrp.p, (int32_t*)mc.p, (double*)mv.p);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(e->stream));
		if (rowptr) HIP_TRY(hipMemcpy(rowptr, mrp.p, sizeof(int64_t) * (size_t)(A.nrows + 1), hipMemcpyDeviceToHost));
		if (colind) HIP_TRY(hipMemcpy(colind, mc.p, sizeof(int32_t) * (size_t)nnz_all, hipMemcpyDeviceToHost));
		if (values) HIP_TRY(hipMemcpy(values, mv.p, e->esz * (size_t)nnz_all, hipMemcpyDeviceToHost));
		return LPP_OK;
	}
	if (rowptr) HIP_TRY(hipMemcpy(rowptr, A.rowptr, sizeof(int64_t) * (size_t)(A.nrows + 1), hipMemcpyDeviceToHost));
	if ((colind || values) && A.nnz) {
		DevScratch scol, sval;
		const int32_t* dcol = nullptr;
		const void* dval = nullptr;
		lpp_status st = plain_of(A, scol, sval, dcol, dval);
		if (st != LPP_OK) return st;
		if (colind) HIP_TRY(hipMemcpy(colind, dcol, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToHost));
		if (values) HIP_TRY(hipMemcpy(values, dval, e->esz * (size_t)A.nnz, hipMemcpyDeviceToHost));
	}
	return LPP_OK;
}

lpp_status lpp_engine_spmv_acc(lpp_engine* e, void* x_inout, const void* y)
{
	if (!e || !x_inout || !y) return fail(LPP_ERR_INVALID, "lpp_engine_spmv_acc: null argument");
	if (!e->has_matrix()) return fail(LPP_ERR_STATE, "lpp_engine_spmv_acc: no matrix (call lpp_engine_set_csr first)");
	if (e->has_comm && e->comm.nranks > 1) return fail(LPP_ERR_STATE, "lpp_engine_spmv_acc: not available on a partitioned matrix");
	if (e->active) return fail(LPP_ERR_STATE, "lpp_engine_spmv_acc: a Lanczos run is active");
	if (e->pb.active && e->pb.tx) return fail(LPP_ERR_STATE, "lpp_engine_spmv_acc: a row slice alone has no product (several GPUs: use the Lanczos entry points)");
	HIP_TRY(hipSetDevice(e->cfg.device));
	lpp_status st = vec_from_host(e, e->x, x_inout);
	if (st != LPP_OK) return st;
	if ((st = vec_from_host(e, e->y, y)) != LPP_OK) return st;
	if (e->pb.active)
		pb_launch(e, e->y, e->x, nullptr);
	else if (e->kron.active)
		kron_launch(e, e->y, e->y, e->x, nullptr);
	else
		spmv_launch(e, e->A_loc, e->y, e->x, nullptr, nullptr);
	HIP_TRY(hipGetLastError());
	if ((st = vec_to_host(e, x_inout, e->x)) != LPP_OK) return st;
	HIP_TRY(hipStreamSynchronize(e->stream));
	return LPP_OK;
}

lpp_status lpp_engine_bench_spmv(lpp_engine* e, int32_t warmup, int32_t iters, double* ms_per_launch)
{
	if (!e || iters <= 0 || !ms_per_launch) return fail(LPP_ERR_INVALID, "lpp_engine_bench_spmv: bad argument");
	if (!e->has_matrix()) return fail(LPP_ERR_STATE, "lpp_engine_bench_spmv: no matrix");
	if (e->active) return fail(LPP_ERR_STATE, "lpp_engine_bench_spmv: a Lanczos run is active");
	if (e->has_comm && e->comm.nranks > 1 && e->tx) return fail(LPP_ERR_STATE, "lpp_engine_bench_spmv: not available with the transposition exchange");
	HIP_TRY(hipSetDevice(e->cfg.device));
	vec_fill_random(e, e->y, 99);
	HIP_TRY(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
	const void* src = e->y;
	if (e->has_comm && e->comm.nranks > 1) {
		// bench the two local kernels against a synthetic gathered vector (no collective in the loop)
		k_fill_random<<<1024, 256, 0, e->stream>>>((double*)e->comm.gath_buf, e->comm.shard_stride * e->comm.nranks * (e->is_complex ? 2 : 1), 0, 98);
	}
	for (int i = 0; i < warmup + iters; i++) {
		if (i == warmup) HIP_TRY(hipEventRecord(e->ev_t0, e->stream));
		if (e->pb.active) {
			pb_launch(e, e->y, e->x, e->partial);
			continue;
		}
		if (e->kron.active) {
			kron_launch(e, e->y, (e->has_comm && e->comm.nranks > 1) ? e->comm.gath_buf : e->y, e->x, e->partial);
			continue;
		}
		spmv_launch(e, e->A_loc, src, e->x, e->y, e->A_rem.nnz ? nullptr : e->partial);
		if (e->A_rem.nnz) spmv_launch(e, e->A_rem, e->comm.gath_buf, e->x, e->y, e->partial);
	}
	HIP_TRY(hipEventRecord(e->ev_t1, e->stream));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventSynchronize(e->ev_t1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e->ev_t0, e->ev_t1));
	*ms_per_launch = (double)ms / iters;
	HIP_TRY(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
	HIP_TRY(hipMemsetAsync(e->y, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return LPP_OK;
}

lpp_status lpp_engine_sync(lpp_engine* e)
{
	if (!e) return fail(LPP_ERR_INVALID, "lpp_engine_sync: null engine");
	HIP_TRY(hipSetDevice(e->cfg.device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return LPP_OK;
}

lpp_status lpp_engine_get_layout(lpp_engine* e, int32_t which, lpp_layout* out)
{
	if (!e || !out || which < 0 || which > 1) return fail(LPP_ERR_INVALID, "lpp_engine_get_layout: bad argument");
	if (e->kron.active) return fail(LPP_ERR_STATE, "lpp_engine_get_layout: the matrix-free engine stores no CSR");
	const DevCsr& A = which == 0 ? e->A_loc : e->A_rem;
	const size_t s = e->esz;
	lpp_layout L {};
	if (e->tj.active && which == 0) {
		const TjState& S = e->tj;
		L.kernel = LPP_SPMV_HOLE_MAJOR;
		L.nnz = S.nnz;
		L.rows_per_block = S.ns;
		L.pieces = 1;
		L.diagonal_plain = 1;
		// the tables (L2 / LDS resident during a product), one f64 of diagonal per stored position, the boundary's permutation
		L.resident_bytes = S.table_bytes + (int64_t)sizeof(double) * S.nblk * S.pitch + (int64_t)sizeof(int32_t) * S.nblk * S.ns;
		L.stream_bytes = (int64_t)sizeof(double) * S.nblk * S.pitch;
		*out = L;
		return LPP_OK;
	}
	if (e->pb.active && which == 0) {
		const PbState& B = e->pb;
		L.kernel = LPP_SPMV_PRODUCT;
		L.coded = 1;
		L.local16 = 1;
		L.block_template = 2;
		L.diagonal_codes = B.dval ? 0 : 1;
		L.shared_stride = B.rowcap;
		L.nnz = B.nnz;
		L.per_row_entries = B.t_entries * B.n_blk; // in-block entries, stored once (the template)
		L.shared_entries = B.c_nnz; // block couplings, stored once per block
		L.rows_per_block = B.n_up;
		L.pieces = B.npieces;
		L.coupling_parts = B.parts ? B.nparts : 1;
		L.chained_step = pb_chain_ok(e) ? 1 : 0;
		L.rows_by_list_length = e->pb.perm && !B.seg ? 1 : 0;
		L.segments = B.seg ? B.seg_nsegs : 0;
		L.coupling_rounds = B.c_nnz > 0 && !B.parts ? B.down_rounds : 1;
		L.diagonal_plain = B.dval ? 1 : 0;
		const size_t small = sizeof(uint32_t) * (size_t)B.f_words + sizeof(uint32_t) * (size_t)B.tw_words + (sizeof(int32_t) + sizeof(uint16_t)) * (size_t)B.spb * (size_t)B.G
		    + (size_t)B.seg_bytes + (size_t)B.t_entries * 12 + (B.chain_model ? 0 : sizeof(int64_t) * (size_t)(B.n_up + 1)) + (size_t)B.c_nnz * 5 + sizeof(int64_t) * 2 * (size_t)(B.n_blk + 1) + 256 * sizeof(double);
		// one diagonal code per row this rank holds, or one plain double when the diagonal has more than 256 distinct values
		const size_t codes = (size_t)(B.tx ? B.nblk_loc : B.n_blk) * (size_t)B.pitch * (B.dval ? 9 : 1);
		L.resident_bytes = (int64_t)(small + codes);
		// per product: one diagonal code per row; the template words and the couplings are re-read from L2 / LDS
		L.stream_bytes = (int64_t)(codes + sizeof(uint32_t) * (size_t)(B.tw_words + B.f_words) + (size_t)B.seg_bytes + (size_t)B.c_nnz * 5);
		*out = L;
		return LPP_OK;
	}
	L.kernel = A.sliced ? (A.window ? LPP_SPMV_WINDOW : LPP_SPMV_SLICED) : LPP_SPMV_ROWGROUP;
	L.coded = A.coded ? 1 : 0;
	L.local16 = A.local16 ? 1 : 0;
	L.block_template = A.tmpl;
	L.diagonal_codes = A.dcode ? 1 : 0;
	L.shared_stride = A.dia_stride;
	L.nnz = A.nnz;
	L.per_row_entries = A.rrowptr ? A.rnnz : A.nnz;
	L.shared_entries = A.ndia;
	L.rows_per_block = A.sliced ? A.geom.B : 0;
	size_t bytes = A.rowptr ? sizeof(int64_t) * (size_t)(A.nrows + 1) : 0;
	size_t stream = 0; // what one product must read of the matrix (see lpp_layout::stream_bytes)
	if (A.col) bytes += sizeof(int32_t) * (size_t)A.nnz;
	if (A.val) bytes += s * (size_t)A.nnz;
	if (A.sliced) {
		const size_t nz = (size_t)L.per_row_entries;
		const size_t struct_slices = A.tmpl ? (size_t)A.geom.spb : (size_t)A.geom.nslices; // block-periodic: block 0 only
		const size_t struct_rows = A.tmpl ? (size_t)A.geom.B : (size_t)A.nrows;
		const size_t struct_nz = A.tmpl ? nz / (size_t)A.geom.nblocks : nz;
		const size_t b_struct = sizeof(int64_t) * (struct_slices + 1) + sizeof(int32_t) * struct_rows // slice_ptr, row_len
		    + (A.local16 ? sizeof(uint16_t) : sizeof(int32_t)) * (struct_nz + 64);
		bytes += b_struct;
		size_t b_val;
		if (A.coded)
			b_val = sizeof(int64_t) * ((A.tmpl == 2 ? (size_t)A.geom.spb : (size_t)A.geom.nslices) + 1) + 256 * sizeof(double) + sizeof(uint32_t) * (size_t)A.code_words;
		else
			b_val = s * (nz + 64);
		bytes += b_val;
		stream += b_struct + b_val; // a template (block 0 only) is counted once: it stays in L2 after the first block
		if (A.rrowptr) {
			const size_t lists = (size_t)A.geom.nslices * (size_t)A.dia_stride * (sizeof(int32_t) + s);
			bytes += sizeof(int64_t) * (size_t)(A.nrows + 1) + lists;
			stream += lists;
		}
		if (A.dcode) {
			bytes += (size_t)A.nrows * (s / 8);
			stream += (size_t)A.nrows * (s / 8);
		}
	} else {
		stream = bytes; // row-group kernel: row pointers, columns and values are all read
	}
	if (A.out_part) { // split-panel layout: the leaving entries as further (sliced, panel-major) matrices + the row map
		for (int q = 0; q < A.split_parts; q++) {
			const DevCsr& O = A.out_part[q];
			const size_t nz = (size_t)O.nnz;
			size_t b = sizeof(int64_t) * (size_t)(O.geom.nslices + 1) + sizeof(int32_t) * (size_t)O.nrows + sizeof(int32_t) * (nz + 64) + sizeof(int32_t) * (size_t)O.nrows;
			b += O.coded ? sizeof(int64_t) * (size_t)(O.geom.nslices + 1) + 256 * sizeof(double) + sizeof(uint32_t) * (size_t)O.code_words : s * (nz + 64);
			stream += b;
			bytes += b + sizeof(int64_t) * (size_t)(O.nrows + 1);
			L.nnz += O.nnz;
			L.per_row_entries += O.nnz;
		}
		L.split_panel = A.split_parts;
	}
	L.stream_bytes = (int64_t)stream;
	L.resident_bytes = (int64_t)bytes;
	*out = L;
	return LPP_OK;
}

lpp_status lpp_engine_get_stats(lpp_engine* e, lpp_stats* s)
{
	if (!e || !s) return fail(LPP_ERR_INVALID, "lpp_engine_get_stats: null argument");
	e->collect_spmv_times();
	*s = e->stats;
	s->nrows = e->n_local;
	s->nnz = e->tj.active ? e->tj.nnz : e->pb.active ? e->pb.nnz_loc : (e->kron.active ? (int64_t)e->kron.equiv_nnz : e->A_loc.nnz + e->A_loc.out_nnz() + e->A_rem.nnz);
	s->spmv_bytes = e->spmv_bytes;
	return LPP_OK;
}

} // extern "C"
