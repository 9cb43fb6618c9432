// lpp_kron_kernels.h -- matrix-free Hubbard product  x += H y  (SURVEY 8(f) N1).
//
// GPU counterpart of the reference's on-the-fly plug-in: InternalProductOnTheFly::matrixVectorProduct
// (src/Engine/InternalProductOnTheFly.h:120-123) -> HubbardHelper::matrixVectorProduct
// (src/Models/HubbardOneOrbital/HubbardHelper.h:105-134).  In the BasisHubbardLanczos ordering
// index = rank(up) + rank(down)*N_up (BasisHubbardLanczos.h:59-63) and with no cross-species sign
// (HubbardHelper.h:214-243) the Hamiltonian is exactly
//     H = H_up (x) 1  +  1 (x) H_down  +  diag( sum_i U_i n_i,up n_i,down ),
// H_s = one-species hopping matrix (+ its potential diagonal) of dimension C(L, n_s).  Only the two
// one-species matrices are stored (a few MB, L2-resident), so an SpMV moves the vectors only:
// ~3.5 GB + 23 GB of coalesced down-hop reads instead of 75 GB of CSR at 4x4 half filling.
//
// One 1024-thread workgroup owns one down-configuration (a block of N_up consecutive rows) at a time:
//   * the block's source entries are staged in LDS (same window as k_spmv_window) -- every up-hop
//     gather is a ds_read;
//   * lane = row: the up part walks the sliced H_up (shared by all blocks), the down part walks the
//     block's H_down row (wave-uniform column, coalesced 512-byte reads of y[jd*N_up + iu]);
//   * the Hubbard-U diagonal comes from popcounts of the basis words.
#pragma once
#include "lpp_kernels.h"

namespace lpp {

constexpr int kKronDownCap = 128; // H_down row entries cached in LDS per block

template <typename T> struct KronArgs {
	SlicedArgs<T> up; // sliced H_up (single block of N_up rows); src / x / ydot are set per block by the kernel
	int64_t n_up;
	int64_t id0, nid; // this rank's first down index (global) and number of down indices
	const int64_t* dn_rowptr; // H_down: plain CSR over global down indices
	const int32_t* dn_col;
	const T* dn_val;
	const uint32_t* up_words; // basis words (L <= 31)
	const uint32_t* dn_words;
	const double* U;
	// Coulomb term of HubbardOneBandExtended (null: none): 0.5 sum_ij V_ij n_i n_j = cdiag_up[iu] + cdiag_dn[id] + sum_{i in up} cross[id][i]
	const double* cdiag_up;
	const double* cdiag_dn;
	const double* cross; // [N_down][32]: (V n_down)_i
	int L;
	const T* ywin; // local slice of y: (id-id0)*N_up + iu
	const T* ydown; // y indexed globally (jd*N_up + iu): the gathered vector on several GPUs, == ywin on one
	T* x; // local slice
	double* partial;
	int xcd_map;
	EpiScale sc;
};

template <typename T, bool DOT, bool WINDOW, bool CODED, int U>
__global__ __launch_bounds__(kWinThreads) void k_spmv_kron(KronArgs<T> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	T* lds = (T*)lds_raw;
	__shared__ double smem[kWinThreads / 64];
	__shared__ double dict_s[CODED ? 256 : 1];
	__shared__ double U_s[32];
	__shared__ double X_s[32];
	__shared__ int32_t dcol_s[kKronDownCap];
	__shared__ T dval_s[kKronDownCap];
	load_dict<CODED>(dict_s, a.up.dict);
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	if (threadIdx.x < 32) U_s[threadIdx.x] = (int)threadIdx.x < a.L ? a.U[threadIdx.x] : 0.0;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int spb = a.up.g.spb;
	int64_t b_begin, b_end, b_stride;
	if (a.xcd_map && (gridDim.x & 7) == 0) {
		const int64_t chunk = (a.nid + 7) / 8;
		const int xcd = blockIdx.x & 7;
		b_begin = xcd * chunk + (blockIdx.x >> 3);
		b_end = min((int64_t)(xcd + 1) * chunk, a.nid);
		b_stride = gridDim.x >> 3;
	} else {
		b_begin = blockIdx.x;
		b_end = a.nid;
		b_stride = gridDim.x;
	}
	double dot = 0.0;
	for (int64_t blk = b_begin; blk < b_end; blk += b_stride) {
		const int64_t gid = a.id0 + blk;
		const T* yblk = a.ywin + blk * a.n_up;
		T* xblk = a.x + blk * a.n_up;
		const int64_t p0 = a.dn_rowptr[gid];
		const int ndn = (int)(a.dn_rowptr[gid + 1] - p0);
		const uint32_t dnw = a.dn_words[gid];
		const double cdn = a.cross ? a.cdiag_dn[gid] : 0.0;
		__syncthreads(); // previous block fully consumed (window, H_down row)
		if (a.cross && threadIdx.x < 32) X_s[threadIdx.x] = a.cross[gid * 32 + threadIdx.x];
		if (WINDOW) {
			for (int64_t i0 = threadIdx.x; i0 < a.n_up; i0 += 8 * kWinThreads) {
				T t[8];
#pragma unroll
				for (int q = 0; q < 8; q++) t[q] = yblk[min(i0 + (int64_t)q * kWinThreads, a.n_up - 1)];
#pragma unroll
				for (int q = 0; q < 8; q++)
					if (i0 + (int64_t)q * kWinThreads < a.n_up) lds[i0 + (int64_t)q * kWinThreads] = t[q];
			}
		}
		if ((int)threadIdx.x < kKronDownCap) { // entries beyond the row are zero-valued and point at the own block
			const bool in = (int)threadIdx.x < ndn;
			dcol_s[threadIdx.x] = in ? a.dn_col[p0 + threadIdx.x] : (int32_t)gid;
			dval_s[threadIdx.x] = in ? a.dn_val[p0 + threadIdx.x] : VT<T>::zero();
		}
		__syncthreads();
		SlicedArgs<T> ua = a.up;
		ua.src = yblk;
		int64_t row0 = 0, base = 0, cbase = 0;
		int nvalid = 0, len = 0;
		if (wave < spb) slice_meta<T, CODED>(ua, wave, row0, nvalid, len, base, cbase);
		for (int j = wave; j < spb; j += kWinThreads / 64) {
			int64_t row0n = 0, basen = 0, cbasen = 0;
			int nvalidn = 0, lenn = 0;
			if (j + kWinThreads / 64 < spb) slice_meta<T, CODED>(ua, j + kWinThreads / 64, row0n, nvalidn, lenn, basen, cbasen);
			if (nvalid > 0) {
				const bool valid = lane < nvalid;
				const int64_t iu = row0 + (valid ? lane : 0);
				const T xold = xblk[iu];
				const uint32_t upw = a.up_words[iu];
				// down part first: wave-uniform column, coalesced reads of y[jd*N_up + iu].  The cached H_down row is
				// padded with zero-valued entries to a multiple of 8; groups of 8 loads are double-buffered so
				// up to 16 independent 512-byte reads are in flight per wave.
				T acc = VT<T>::zero();
				const int ngroups = (min(ndn, kKronDownCap) + 7) >> 3;
				T g0[8], g1[8];
				if (ngroups > 0) {
#pragma unroll
					for (int q = 0; q < 8; q++) g0[q] = a.ydown[(int64_t)dcol_s[q] * a.n_up + iu];
				}
				for (int gk = 0; gk < ngroups; gk++) {
					if (gk + 1 < ngroups) {
#pragma unroll
						for (int q = 0; q < 8; q++) g1[q] = a.ydown[(int64_t)dcol_s[(gk + 1) * 8 + q] * a.n_up + iu];
					}
#pragma unroll
					for (int q = 0; q < 8; q++) VT<T>::mac(acc, dval_s[gk * 8 + q], g0[q]);
#pragma unroll
					for (int q = 0; q < 8; q++) g0[q] = g1[q];
				}
				for (int p = kKronDownCap; p < ndn; p++) VT<T>::mac(acc, a.dn_val[p0 + p], a.ydown[(int64_t)a.dn_col[p0 + p] * a.n_up + iu]);
				// up part: sliced H_up, gathers from the LDS window (or the L2-resident block)
				const T accu = sliced_accumulate<T, WINDOW, CODED, U>(ua, len, base, cbase, lds, 0, (uint32_t)a.n_up, dict_s, (int32_t)iu);
				acc = VT<T>::add(acc, accu);
				// Hubbard U on the doubly occupied sites
				const T yc = WINDOW ? lds[iu] : yblk[iu];
				double ud = 0.0;
				for (uint32_t m = upw & dnw; m; m &= m - 1) ud += U_s[__ffs((int)m) - 1];
				if (a.cross) {
					ud += a.cdiag_up[iu] + cdn;
					for (uint32_t m = upw; m; m &= m - 1) ud += X_s[__ffs((int)m) - 1];
				}
				T t = VT<T>::zero();
				if (sizeof(T) == 16) {
					cplx* tc = (cplx*)&t;
					const cplx* yy = (const cplx*)&yc;
					tc->re = ud * yy->re;
					tc->im = ud * yy->im;
				} else {
					*(double*)&t = ud * *(const double*)&yc;
				}
				acc = VT<T>::add(acc, t);
				if (valid) {
					const T xv = epi_lin(beta, xold, alpha, acc);
					xblk[iu] = xv;
					if (DOT) dot += VT<T>::dot_re(yc, xv);
				}
			}
			row0 = row0n;
			base = basen;
			cbase = cbasen;
			nvalid = nvalidn;
			len = lenn;
		}
	}
	if (DOT) {
		const double r = block_sum_n<kWinThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// Packed variant (the common case: <= 256 distinct matrix values; N_up < 2^24 real, <= 65536 complex).
// H_up is tiny (a few hundred KB), so it is stored PADDED, slice-major [slice][slot][lane], one 32-bit
// word per entry:  real: column (24 bit) | code << 24;  complex: column (16 bit) | code_re << 16 | code_im << 24.  Padding entries carry
// the code of 0.0 and the row's own column.  Per up-hop the inner loop is: one coalesced 4-byte load
// (L2 hit), two LDS reads (dictionary, window) and an FMA -- no ballot/popcount compaction, no 64-bit
// address arithmetic (measured: the compact sliced walk cost ~27 VALU instructions per entry).
// The down part reads the block's H_down row from LDS as precomputed 64-bit element offsets jd*N_up.
// ---------------------------------------------------------------------------------------------
template <typename T> struct KronPackedArgs {
	const uint32_t* words; // packed H_up
	const int32_t* slice_off; // first word of slice j (in words), spb+1 entries
	const int32_t* slice_len; // padded slots of slice j
	const double* dict; // 256 doubles
	int spb;
	int64_t n_up;
	int64_t id0, nid;
	const int64_t* dn_rowptr;
	const int32_t* dn_col;
	const T* dn_val;
	const uint32_t* up_words;
	const uint32_t* dn_words;
	const double* U;
	const double *cdiag_up, *cdiag_dn, *cross; // Coulomb term (see KronArgs), null: none
	int L;
	const T* ywin;
	const T* ydown;
	T* x;
	double* partial;
	int xcd_map;
	EpiScale sc;
	// 0: whole product.  Transposition exchange: 1 = up-hops + U diagonal only (own slice); 2 = down-hops only, on the
	// transposed slice (n_up = rows per down index = peru, id0 = 0, nid = padded number of down indices, n_dn valid)
	int part;
	int64_t n_dn;
};

template <typename T> __device__ __forceinline__ T kron_decode(uint32_t w, const double* dict);
template <> __device__ __forceinline__ double kron_decode<double>(uint32_t w, const double* dict) { return dict[w >> 24]; }
template <typename T> __device__ __forceinline__ uint32_t kron_col(uint32_t w) { return sizeof(T) == 16 ? (w & 0xffffu) : (w & 0xffffffu); }
template <> __device__ __forceinline__ cplx kron_decode<cplx>(uint32_t w, const double* dict)
{
	return cplx { dict[(w >> 16) & 0xffu], dict[w >> 24] };
}

template <typename T, bool DOT, bool WINDOW>
__global__ __launch_bounds__(kWinThreads) void k_spmv_kron_packed(KronPackedArgs<T> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	T* lds = (T*)lds_raw;
	__shared__ double smem[kWinThreads / 64];
	__shared__ double dict_s[256];
	__shared__ double U_s[32];
	__shared__ double X_s[32];
	__shared__ long long doff_s[kKronDownCap]; // jd * N_up
	__shared__ T dval_s[kKronDownCap];
	__shared__ int next_slice;
	for (int i = threadIdx.x; i < 256; i += kWinThreads) dict_s[i] = a.dict[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	if (threadIdx.x < 32) U_s[threadIdx.x] = (int)threadIdx.x < a.L ? a.U[threadIdx.x] : 0.0;
	const int lane = threadIdx.x & 63;
	int64_t b_begin, b_end, b_stride;
	if (a.xcd_map && (gridDim.x & 7) == 0) {
		const int64_t chunk = (a.nid + 7) / 8;
		const int xcd = blockIdx.x & 7;
		b_begin = xcd * chunk + (blockIdx.x >> 3);
		b_end = min((int64_t)(xcd + 1) * chunk, a.nid);
		b_stride = gridDim.x >> 3;
	} else {
		b_begin = blockIdx.x;
		b_end = a.nid;
		b_stride = gridDim.x;
	}
	double dot = 0.0;
	for (int64_t blk = b_begin; blk < b_end; blk += b_stride) {
		const int64_t gid = a.id0 + blk;
		if (a.part == 2 && gid >= a.n_dn) continue; // padded down index of the transposed layout (uniform per workgroup)
		const T* yblk = a.ywin + blk * a.n_up;
		T* xblk = a.x + blk * a.n_up;
		const int64_t p0 = a.dn_rowptr[gid];
		const int ndn = (int)(a.dn_rowptr[gid + 1] - p0);
		const uint32_t dnw = a.dn_words[gid];
		const bool coul = a.cross != nullptr && a.part != 2;
		const double cdn = coul ? a.cdiag_dn[gid] : 0.0;
		__syncthreads(); // previous block fully consumed (window, H_down row)
		if (threadIdx.x == 0) next_slice = 0;
		if (coul && threadIdx.x < 32) X_s[threadIdx.x] = a.cross[gid * 32 + threadIdx.x];
		if (WINDOW && a.part != 2) {
			for (int64_t i0 = threadIdx.x; i0 < a.n_up; i0 += 8 * kWinThreads) {
				T t[8];
#pragma unroll
				for (int q = 0; q < 8; q++) t[q] = yblk[min(i0 + (int64_t)q * kWinThreads, a.n_up - 1)];
#pragma unroll
				for (int q = 0; q < 8; q++)
					if (i0 + (int64_t)q * kWinThreads < a.n_up) lds[i0 + (int64_t)q * kWinThreads] = t[q];
			}
		}
		if ((int)threadIdx.x < kKronDownCap) { // entries beyond the row are zero-valued and point at the own block
			const bool in = (int)threadIdx.x < ndn;
			doff_s[threadIdx.x] = (long long)(in ? a.dn_col[p0 + threadIdx.x] : (int32_t)gid) * a.n_up;
			dval_s[threadIdx.x] = in ? a.dn_val[p0 + threadIdx.x] : VT<T>::zero();
		}
		__syncthreads();
		const int ngroups = (a.part == 1) ? 0 : ((min(ndn, kKronDownCap) + 7) >> 3);
		// slices are claimed dynamically (see k_spmv_window)
		for (int j = next_slice_claim(&next_slice); j < a.spb; j = next_slice_claim(&next_slice)) {
			const int iu_raw = j * 64 + lane;
			const bool valid = iu_raw < a.n_up;
			const int iu = valid ? iu_raw : (int)a.n_up - 1;
			const T xold = xblk[iu];
			const uint32_t upw = (a.part == 2) ? 0u : a.up_words[iu];
			const uint32_t* wp = a.words + a.slice_off[j] + lane;
			const int ml = (a.part == 2) ? 0 : a.slice_len[j]; // multiple of 8
			// first batch of H_up words and of down reads are requested together
			uint32_t w0[8], w1[8];
			if (ml > 0) {
#pragma unroll
				for (int q = 0; q < 8; q++) w0[q] = wp[q * 64];
			}
			T acc = VT<T>::zero();
			T g0[8], g1[8];
			const T* yd = a.ydown + iu;
			if (ngroups > 0) {
#pragma unroll
				for (int q = 0; q < 8; q++) g0[q] = yd[doff_s[q]];
			}
			for (int gk = 0; gk < ngroups; gk++) {
				if (gk + 1 < ngroups) {
#pragma unroll
					for (int q = 0; q < 8; q++) g1[q] = yd[doff_s[(gk + 1) * 8 + q]];
				}
#pragma unroll
				for (int q = 0; q < 8; q++) VT<T>::mac(acc, dval_s[gk * 8 + q], g0[q]);
#pragma unroll
				for (int q = 0; q < 8; q++) g0[q] = g1[q];
			}
			if (a.part != 1)
				for (int p = kKronDownCap; p < ndn; p++) VT<T>::mac(acc, a.dn_val[p0 + p], yd[(int64_t)a.dn_col[p0 + p] * a.n_up]);
			// up part
			for (int k = 0; k < ml; k += 8) {
				if (k + 8 < ml) {
#pragma unroll
					for (int q = 0; q < 8; q++) w1[q] = wp[(k + 8 + q) * 64];
				}
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const uint32_t c = kron_col<T>(w0[q]);
					const T g = WINDOW ? lds[c] : yblk[c];
					VT<T>::mac(acc, kron_decode<T>(w0[q], dict_s), g);
				}
#pragma unroll
				for (int q = 0; q < 8; q++) w0[q] = w1[q];
			}
			// Hubbard U on the doubly occupied sites
			const T yc = (a.part == 2) ? VT<T>::zero() : (WINDOW ? lds[iu] : yblk[iu]);
			double ud = 0.0;
			for (uint32_t m = upw & dnw; m; m &= m - 1) ud += U_s[__ffs((int)m) - 1];
			if (coul) {
				ud += a.cdiag_up[iu] + cdn;
				for (uint32_t m = upw; m; m &= m - 1) ud += X_s[__ffs((int)m) - 1];
			}
			T t = VT<T>::zero();
			if (sizeof(T) == 16) {
				cplx* tc = (cplx*)&t;
				const cplx* yy = (const cplx*)&yc;
				tc->re = ud * yy->re;
				tc->im = ud * yy->im;
			} else {
				*(double*)&t = ud * *(const double*)&yc;
			}
			acc = VT<T>::add(acc, t);
			if (valid) {
				const T xv = epi_lin(beta, xold, alpha, acc);
				xblk[iu] = xv;
				if (DOT) dot += VT<T>::dot_re(yc, xv);
			}
		}
	}
	if (DOT) {
		const double r = block_sum_n<kWinThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// Chunked-window variant for one-species spaces that exceed LDS (3x6 lattice: N_up = 48620 doubles = 389 KB).
// Without a window every up-hop is a scattered 8-byte global gather, i.e. 64 L1 tag accesses per wave instruction
// (measured: the L1, not L2 or HBM, bounds that shape), 80 of the 131 ms of a 2.36e9-state product.  Here the block's
// source row y[id][:] is staged in nchunk pieces of `cw` columns; H_up is packed once per piece (entries whose column
// lies in the piece, columns relative to it), and pass c adds the piece's contribution to x: pass 0 also applies the
// down-hops, the U diagonal and beta*x_old.  x is re-read between passes, but one block of x (389 KB) stays in L2.
// ---------------------------------------------------------------------------------------------
template <typename T> struct KronChunkArgs {
	KronPackedArgs<T> k; // words / slice_off / slice_len hold nchunk consecutive packings: off[c*(spb+1)+j], len[c*spb+j]
	int nchunk;
	int cw; // columns per piece (multiple of 64)
};

template <typename T, bool DOT>
__global__ __launch_bounds__(kWinThreads) void k_spmv_kron_chunked(KronChunkArgs<T> ca)
{
	const KronPackedArgs<T>& a = ca.k;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	T* lds = (T*)lds_raw;
	__shared__ double smem[kWinThreads / 64];
	__shared__ double dict_s[256];
	__shared__ double U_s[32];
	__shared__ double X_s[32];
	__shared__ long long doff_s[kKronDownCap];
	__shared__ T dval_s[kKronDownCap];
	__shared__ int next_slice;
	for (int i = threadIdx.x; i < 256; i += kWinThreads) dict_s[i] = a.dict[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	if (threadIdx.x < 32) U_s[threadIdx.x] = (int)threadIdx.x < a.L ? a.U[threadIdx.x] : 0.0;
	const int lane = threadIdx.x & 63;
	double dot = 0.0;
	for (int64_t blk = blockIdx.x; blk < a.nid; blk += gridDim.x) {
		const int64_t gid = a.id0 + blk;
		const T* yblk = a.ywin + blk * a.n_up;
		T* xblk = a.x + blk * a.n_up;
		const int64_t p0 = a.dn_rowptr[gid];
		const int ndn = (int)(a.dn_rowptr[gid + 1] - p0);
		const uint32_t dnw = a.dn_words[gid];
		for (int c = 0; c < ca.nchunk; c++) {
			const int64_t c0 = (int64_t)c * ca.cw, c1 = min(c0 + ca.cw, a.n_up);
			__syncthreads(); // previous pass fully consumed (window piece, H_down row)
			if (threadIdx.x == 0) next_slice = 0;
			for (int64_t i0 = c0 + threadIdx.x; i0 < c1; i0 += 8 * kWinThreads) {
				T t[8];
#pragma unroll
				for (int q = 0; q < 8; q++) t[q] = yblk[min(i0 + (int64_t)q * kWinThreads, c1 - 1)];
#pragma unroll
				for (int q = 0; q < 8; q++)
					if (i0 + (int64_t)q * kWinThreads < c1) lds[i0 - c0 + (int64_t)q * kWinThreads] = t[q];
			}
			if (c == 0 && a.cross && threadIdx.x < 32) X_s[threadIdx.x] = a.cross[gid * 32 + threadIdx.x];
			if (c == 0 && (int)threadIdx.x < kKronDownCap) {
				const bool in = (int)threadIdx.x < ndn;
				doff_s[threadIdx.x] = (long long)(in ? a.dn_col[p0 + threadIdx.x] : (int32_t)gid) * a.n_up;
				dval_s[threadIdx.x] = in ? a.dn_val[p0 + threadIdx.x] : VT<T>::zero();
			}
			__syncthreads();
			// part == 1 (transposition exchange): up-hops + U diagonal only, the down-hops act on the transposed slice elsewhere
			const int ngroups = (c == 0 && a.part != 1) ? ((min(ndn, kKronDownCap) + 7) >> 3) : 0;
			const int32_t* offc = a.slice_off + (int64_t)c * (a.spb + 1);
			const int32_t* lenc = a.slice_len + (int64_t)c * a.spb;
			for (int j = next_slice_claim(&next_slice); j < a.spb; j = next_slice_claim(&next_slice)) {
				const int iu_raw = j * 64 + lane;
				const bool valid = iu_raw < a.n_up;
				const int iu = valid ? iu_raw : (int)a.n_up - 1;
				const T xold = xblk[iu];
				const uint32_t* wp = a.words + offc[j] + lane;
				const int ml = lenc[j]; // multiple of 8
				uint32_t w0[8], w1[8];
				if (ml > 0) {
#pragma unroll
					for (int q = 0; q < 8; q++) w0[q] = wp[q * 64];
				}
				T acc = VT<T>::zero();
				T yc = VT<T>::zero();
				if (c == 0 || (DOT && c == ca.nchunk - 1)) yc = yblk[iu];
				if (c == 0) {
					T g0[8], g1[8];
					const T* yd = a.ydown + iu;
					if (ngroups > 0) {
#pragma unroll
						for (int q = 0; q < 8; q++) g0[q] = yd[doff_s[q]];
					}
					for (int gk = 0; gk < ngroups; gk++) {
						if (gk + 1 < ngroups) {
#pragma unroll
							for (int q = 0; q < 8; q++) g1[q] = yd[doff_s[(gk + 1) * 8 + q]];
						}
#pragma unroll
						for (int q = 0; q < 8; q++) VT<T>::mac(acc, dval_s[gk * 8 + q], g0[q]);
#pragma unroll
						for (int q = 0; q < 8; q++) g0[q] = g1[q];
					}
					if (a.part != 1)
						for (int p = kKronDownCap; p < ndn; p++) VT<T>::mac(acc, a.dn_val[p0 + p], yd[(int64_t)a.dn_col[p0 + p] * a.n_up]);
					// Hubbard U on the doubly occupied sites
					double ud = 0.0;
					const uint32_t upw = a.up_words[iu];
					for (uint32_t m = upw & dnw; m; m &= m - 1) ud += U_s[__ffs((int)m) - 1];
					if (a.cross) { // Coulomb term of HubbardOneBandExtended
						ud += a.cdiag_up[iu] + a.cdiag_dn[gid];
						for (uint32_t m = upw; m; m &= m - 1) ud += X_s[__ffs((int)m) - 1];
					}
					T t = yc;
					double* td = (double*)&t;
					td[0] *= ud;
					if (sizeof(T) == 16) td[1] *= ud;
					acc = VT<T>::add(acc, t);
				}
				for (int k = 0; k < ml; k += 8) {
					if (k + 8 < ml) {
#pragma unroll
						for (int q = 0; q < 8; q++) w1[q] = wp[(k + 8 + q) * 64];
					}
#pragma unroll
					for (int q = 0; q < 8; q++) VT<T>::mac(acc, kron_decode<T>(w0[q], dict_s), lds[kron_col<T>(w0[q])]);
#pragma unroll
					for (int q = 0; q < 8; q++) w0[q] = w1[q];
				}
				if (valid) {
					const T xv = epi_lin(c == 0 ? beta : 1.0, xold, alpha, acc);
					xblk[iu] = xv;
					if (DOT && c == ca.nchunk - 1) dot += VT<T>::dot_re(yc, xv);
				}
			}
		}
	}
	if (DOT) {
		const double r = block_sum_n<kWinThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// basis words of one species: word(i) = i-th L-bit word with n set bits, ascending
static __global__ void k_basis_words(const uint64_t* __restrict__ comb, int combdim, int64_t count, int nbits, int L,
                                     uint32_t* __restrict__ out)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	uint64_t w = 0;
	int64_t r = i;
	int k = nbits;
	for (int b = L - 1; b >= 0 && k > 0; b--) {
		const int64_t c = (int64_t)comb[b * combdim + k];
		if (r >= c) {
			w |= 1ull << b;
			r -= c;
			k--;
		}
	}
	out[i] = (uint32_t)w;
}

} // namespace lpp
