// lpp_assemble_kernels.h -- on-device CSR assembly for the three in-scope models.
//
// A many-body state is one 64-bit word  w = (down << L) | up   (Heisenberg S=1/2: w = spin word).
// In every in-scope basis the reference's state index is MONOTONE in w:
//   Hubbard    index = rank(up) + rank(down)*N_up          BasisHubbardLanczos.h:59-63
//   Heisenberg index = position in the ascending list       BasisHeisenberg.h:38-46 (= rank(w) for S=1/2)
//   t-J        index = position in the sorted list          BasisTjMultiOrbLanczos.h:29-42
// and every off-diagonal term maps ket -> bra = ket ^ xmask with bra - ket = delta a constant of
// the term.  The host therefore sorts the term ("process") list by delta ONCE; a row is produced
// by walking that list in order, which emits the entries already sorted by column exactly as
// SparseRow::finalize would -- no per-row sort, no stored basis, O(L) ranking instead of the
// reference's linear-scan perfectIndex (BasisHeisenberg.h:73-80) with the same resulting index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpp_kernels.h"

namespace lpp {

enum { ASM_HUBBARD = 0, ASM_HEISENBERG = 1, ASM_TJ = 2, ASM_HEISENBERG_S = 3 };

struct Proc {
	uint64_t need_set, need_clear, xmask;
	uint64_t smask_ket, smask_bra; // sign = (-1)^(popc(ket&smask_ket)+popc(bra&smask_bra)+sign_const)
	double amp_re, amp_im;
	int32_t sign_const;
	int32_t real_only; // 1: the imaginary part is the constant +0.0 (term added as a real number by the reference)
};

constexpr int kCombDim = 65;

struct AsmParams {
	int model;
	int L;
	int nup, ndown;
	int nproc; // processes sorted by delta ascending
	int nneg; // processes with delta < 0 (emitted before the diagonal)
	int64_t n_up; // Hubbard: C(L,nup); t-J: C(L-ndown,nup)
	int64_t nrows_global;
	int64_t row0, nloc; // rows [row0, row0+nloc) are produced
	int part; // 0: all columns; 1: columns in [col_lo,col_hi) shifted by -col_lo; 2: columns outside
	int64_t col_lo, col_hi;
	const Proc* procs;
	const uint64_t* comb; // kCombDim x kCombDim binomials
	const double* d0; // Hubbard U[L] | Heisenberg field[L] (nd0 valid) | t-J potentialV[2L]
	const double* d1; // Hubbard V[L] | Heisenberg anisotropy[L] (nd1 valid) | t-J jzz[L*L]
	const double* d2; // Hubbard Coulomb coupling[L*L] or null | Heisenberg jzz[L*L] | t-J w[L*L]
	const double* d3; // Hubbard spin coupling J[L*L] (SuperHubbardExtended) or null
	int nd0, nd1;
	// Hubbard, transposed row layout (multi-GPU transposition scheme): row = id*peru + (iu - iu0) for the rank's
	// UP-index range [iu0, iu0+nu) and ALL down indices id < n_dn; rows with iu_l >= nu or id >= n_dn are padding
	int tr;
	int64_t iu0, nu, peru, n_dn;
	int no_diag; // 1: do not emit the diagonal (the down-hop part of a split matrix)
	// Heisenberg with spin S > 1/2 (ASM_HEISENBERG_S): a state is L digits of `bits` bits, digit = m + S in [0, dmax], digit sum = msum
	// (BasisHeisenberg.h:28-46).  A term (i, j) raises digit i and lowers digit j: Proc.need_set = bit offset of i, Proc.need_clear =
	// bit offset of j; its value depends on digit j only: amp[term * (twiceS + 1) + digit_j] (Heisenberg.h:278-307, formed on the host).
	int twiceS, bits, dmax, msum;
	int sdim; // row length of `digits`
	const uint64_t* digits; // digits[l * sdim + s] = number of l-digit strings with digit sum s
	const double* amp;
};

__device__ __forceinline__ uint64_t comb_at(const uint64_t* comb, int n, int m) { return comb[n * kCombDim + m]; }

// rank in the ascending list of words with fixed popcount (BasisOneSpin::perfectIndex, BasisOneSpin.h:73-81)
__device__ __forceinline__ int64_t rank_comb(const uint64_t* comb, uint64_t state)
{
	int64_t n = 0;
	int c = 1;
	while (state) {
		const int b = __ffsll((unsigned long long)state) - 1;
		n += (int64_t)comb_at(comb, b, c++);
		state &= state - 1;
	}
	return n;
}

__device__ __forceinline__ uint64_t unrank_comb(const uint64_t* comb, int64_t r, int k, int nbits)
{
	uint64_t w = 0;
	for (int b = nbits - 1; b >= 0 && k > 0; b--) {
		const int64_t c = (int64_t)comb_at(comb, b, k);
		if (r >= c) {
			w |= 1ull << b;
			r -= c;
			k--;
		}
	}
	return w;
}

// software pext / pdep over the low L bits
__device__ __forceinline__ uint64_t pext_sw(uint64_t v, uint64_t mask)
{
	uint64_t out = 0;
	int k = 0;
	while (mask) {
		const uint64_t low = mask & (~mask + 1);
		if (v & low) out |= 1ull << k;
		k++;
		mask &= mask - 1;
	}
	return out;
}
__device__ __forceinline__ uint64_t pdep_sw(uint64_t v, uint64_t mask)
{
	uint64_t out = 0;
	while (mask && v) {
		const uint64_t low = mask & (~mask + 1);
		if (v & 1) out |= low;
		v >>= 1;
		mask &= mask - 1;
	}
	return out;
}

// Position of a digit string in the ascending list of all strings with the same digit sum (digits in [0, dmax]); ascending as
// integers = lexicographic from the most significant digit.  BasisHeisenberg::perfectIndex (BasisHeisenberg.h:73-80) finds the
// same position by scanning the stored list.
__device__ __forceinline__ int64_t rank_digits(const AsmParams& P, uint64_t w)
{
	const uint64_t dm = (1ull << P.bits) - 1;
	int srem = P.msum;
	int64_t r = 0;
	for (int p = P.L - 1; p >= 0; p--) {
		const int d = (int)((w >> (p * P.bits)) & dm);
		for (int k = 0; k < d; k++)
			if (srem - k >= 0) r += (int64_t)P.digits[p * P.sdim + (srem - k)];
		srem -= d;
	}
	return r;
}
__device__ __forceinline__ uint64_t unrank_digits(const AsmParams& P, int64_t r)
{
	uint64_t w = 0;
	int srem = P.msum;
	for (int p = P.L - 1; p >= 0; p--) {
		int d = 0;
		for (; d < P.dmax; d++) {
			const int64_t c = (srem - d >= 0) ? (int64_t)P.digits[p * P.sdim + (srem - d)] : 0;
			if (r < c) break;
			r -= c;
		}
		w |= (uint64_t)d << (p * P.bits);
		srem -= d;
	}
	return w;
}

// ket -> bra of one off-diagonal term; false when the term does not apply to this ket
template <int MODEL> __device__ __forceinline__ bool proc_bra(const AsmParams& P, const Proc& pr, uint64_t ket, uint64_t& bra, int& dj)
{
	if (MODEL == ASM_HEISENBERG_S) {
		const uint64_t dm = (1ull << P.bits) - 1;
		const int si = (int)pr.need_set, sj = (int)pr.need_clear;
		const int v1 = (int)((ket >> si) & dm);
		dj = (int)((ket >> sj) & dm);
		if (v1 == P.twiceS || dj == 0) return false; // Heisenberg.h:103 and :294
		bra = ket + (1ull << si) - (1ull << sj);
		return true;
	}
	dj = 0;
	if ((ket & pr.need_set) != pr.need_set || (ket & pr.need_clear) != 0) return false;
	bra = ket ^ pr.xmask;
	return true;
}

template <int MODEL> __device__ __forceinline__ uint64_t state_of(const AsmParams& P, int64_t idx)
{
	const uint64_t lowmask = (P.L >= 64) ? ~0ull : ((1ull << P.L) - 1);
	if (MODEL == ASM_HEISENBERG_S) return unrank_digits(P, idx);
	if (MODEL == ASM_HEISENBERG) return unrank_comb(P.comb, idx, P.nup, P.L);
	const int64_t iu = idx % P.n_up, id = idx / P.n_up;
	const uint64_t down = unrank_comb(P.comb, id, P.ndown, P.L);
	uint64_t up;
	if (MODEL == ASM_HUBBARD)
		up = unrank_comb(P.comb, iu, P.nup, P.L);
	else
		up = pdep_sw(unrank_comb(P.comb, iu, P.nup, P.L - P.ndown), ~down & lowmask);
	return (down << P.L) | up;
}

template <int MODEL> __device__ __forceinline__ int64_t index_of(const AsmParams& P, uint64_t w)
{
	if (MODEL == ASM_HEISENBERG_S) return rank_digits(P, w);
	if (MODEL == ASM_HEISENBERG) return rank_comb(P.comb, w);
	const uint64_t lowmask = (1ull << P.L) - 1;
	const uint64_t up = w & lowmask, down = w >> P.L;
	if (MODEL == ASM_HUBBARD && P.tr) return rank_comb(P.comb, down) * P.peru + (rank_comb(P.comb, up) - P.iu0);
	if (MODEL == ASM_HUBBARD) return rank_comb(P.comb, up) + rank_comb(P.comb, down) * P.n_up;
	return rank_comb(P.comb, pext_sw(up, ~down & lowmask)) + rank_comb(P.comb, down) * P.n_up;
}

// state of (global or transposed-layout) row `row`; false for a padding row of the transposed layout
template <int MODEL> __device__ __forceinline__ bool row_state(const AsmParams& P, int64_t row, uint64_t& ket)
{
	if (MODEL == ASM_HUBBARD && P.tr) {
		const int64_t iul = row % P.peru, id = row / P.peru;
		if (iul >= P.nu || id >= P.n_dn) return false;
		const uint64_t down = unrank_comb(P.comb, id, P.ndown, P.L);
		const uint64_t up = unrank_comb(P.comb, P.iu0 + iul, P.nup, P.L);
		ket = (down << P.L) | up;
		return true;
	}
	ket = state_of<MODEL>(P, row);
	return true;
}

// Diagonal elements, additions in the reference's loop order so the doubles are bit-identical
// (every product below is exact: factors are 0, +-1/2, +-1, 2 or powers of two).
template <int MODEL> __device__ double diag_of(const AsmParams& P, uint64_t w)
{
	const int L = P.L;
	double s = 0.0;
	if (MODEL == ASM_HUBBARD) { // HubbardHelper.h:147-187 (U, Coulomb and potentialV terms, in that order per site)
		const uint64_t up = w & ((1ull << L) - 1), down = w >> L;
		for (int i = 0; i < L; i++) {
			const int nu = (int)((up >> i) & 1), nd = (int)((down >> i) & 1);
			s += P.d0[i] * nu * nd;
			if (P.d3) { // SuperHubbardExtended: sum_j J(i,j) 0.5 Sz_i Sz_j (HubbardHelper.h:158-165); every product is exact
				const double szi = 0.5 * (double)(nu - nd);
				for (int j = 0; j < L; j++) {
					const double value = P.d3[i * L + j];
					if (value == 0) continue;
					const double szj = 0.5 * (double)((int)((up >> j) & 1) - (int)((down >> j) & 1));
					s += value * 0.5 * szi * szj;
				}
			}
			const double ne = nu + nd;
			if (P.d2) { // HubbardOneBandExtended: 0.5 * coulombCoupling(i,j) n_i n_j over ALL j (HubbardHelper.h:167-177)
				for (int j = 0; j < L; j++) {
					const double value = 0.5 * P.d2[i * L + j];
					if (value == 0) continue;
					const double tmp2 = (double)(int)(((up >> j) & 1) + ((down >> j) & 1));
					s += value * ne * tmp2;
				}
			}
			const double tmp = P.d1[i];
			if (tmp != 0) s += tmp * ne;
		}
	} else if (MODEL == ASM_HEISENBERG) { // Heisenberg.h:251-275, twiceS == 1
		for (int i = 0; i < L; i++) {
			const double tmp1 = (double)((w >> i) & 1) - 0.5;
			const double tmp1d = tmp1 * tmp1;
			if (i < P.nd0) s += P.d0[i] * tmp1;
			if (i < P.nd1) s += P.d1[i] * tmp1d;
			for (int j = i + 1; j < L; j++) {
				const double tmp2 = (double)((w >> j) & 1) - 0.5;
				s += tmp1 * tmp2 * P.d2[i * L + j];
			}
		}
	} else if (MODEL == ASM_HEISENBERG_S) {
		// Heisenberg.h:251-275 for any spin.  m_i = digit - S is a multiple of 1/2, so m_i m_j is exact, but its product with a
		// coupling is not: every multiplication and addition must be rounded on its own, as the host compiler does it, for the
		// sum to be the reference's double bit for bit -- so no contraction into fused multiply-adds in this block (measured:
		// with it, 107 of 580 diagonal elements of an S = 3/2 chain are one unit in the last place off).
#pragma clang fp contract(off)
		const uint64_t dm = (1ull << P.bits) - 1;
		const double half = P.twiceS * 0.5;
		for (int i = 0; i < L; i++) {
			const double tmp1 = (double)(int)((w >> (i * P.bits)) & dm) - half;
			const double tmp1d = tmp1 * tmp1;
			if (i < P.nd0) {
				const double t = P.d0[i] * tmp1;
				s = s + t;
			}
			if (i < P.nd1) {
				const double t = P.d1[i] * tmp1d;
				s = s + t;
			}
			for (int j = i + 1; j < L; j++) {
				const double tmp2 = (double)(int)((w >> (j * P.bits)) & dm) - half;
				const double t = tmp1 * tmp2 * P.d2[i * L + j];
				s = s + t;
			}
		}
	} else { // TjMultiOrb.h:597-645, orbitals == 1
		const uint64_t up = w & ((1ull << L) - 1), down = w >> L;
		for (int i = 0; i < L; i++) {
			const int niup = (int)((up >> i) & 1), nidown = (int)((down >> i) & 1);
			if (i < P.nd0) {
				s += P.d0[i] * niup;
				s += P.d0[i + L] * nidown;
			}
			for (int j = i + 1; j < L; j++) {
				const int njup = (int)((up >> j) & 1), njdown = (int)((down >> j) & 1);
				s += (niup - nidown) * (njup - njdown) * P.d1[i * L + j] * 0.25;
				s += (niup + nidown) * (njup + njdown) * P.d2[i * L + j];
			}
		}
	}
	return s;
}

__device__ __forceinline__ bool col_selected(const AsmParams& P, int64_t col)
{
	if (P.part == 0) return true;
	const bool inside = col >= P.col_lo && col < P.col_hi;
	return P.part == 1 ? inside : !inside;
}

// pass 1: entries per row
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_asm_count(AsmParams P, int64_t* __restrict__ len)
{
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const int64_t row = P.row0 + r;
		uint64_t ket;
		if (!row_state<MODEL>(P, row, ket)) {
			len[r] = 0;
			continue;
		}
		int n = 0;
		for (int p = 0; p < P.nproc; p++) {
			const Proc& pr = P.procs[p];
			uint64_t bra;
			int dj;
			if (!proc_bra<MODEL>(P, pr, ket, bra, dj)) continue;
			if (P.part == 0) {
				n++;
			} else {
				const int64_t c = index_of<MODEL>(P, bra);
				if (col_selected(P, c)) n++;
			}
		}
		if (!P.no_diag && col_selected(P, row)) n++; // the diagonal is always stored (HubbardHelper.h:93)
		len[r] = n;
	}
}

// pass 2: fill col / val (len already scanned into rowptr)
template <int MODEL, typename T>
__global__ __launch_bounds__(kBlock) void k_asm_fill(AsmParams P, const int64_t* __restrict__ rowptr,
                                                      int32_t* __restrict__ col, T* __restrict__ val)
{
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const int64_t row = P.row0 + r;
		uint64_t ket;
		if (!row_state<MODEL>(P, row, ket)) continue;
		int64_t q = rowptr[r];
		const int64_t shift = (P.part == 1) ? P.col_lo : 0;
		for (int p = 0; p <= P.nproc; p++) {
			if (p == P.nneg) { // diagonal sits between negative and positive deltas
				if (!P.no_diag && col_selected(P, row)) {
					col[q] = (int32_t)(row - shift);
					T v;
					if constexpr (sizeof(T) == 16) {
						v.re = diag_of<MODEL>(P, ket);
						v.im = 0.0;
					} else {
						v = diag_of<MODEL>(P, ket);
					}
					val[q] = v;
					q++;
				}
			}
			if (p == P.nproc) break;
			const Proc& pr = P.procs[p];
			uint64_t bra;
			int dj;
			if (!proc_bra<MODEL>(P, pr, ket, bra, dj)) continue;
			const int64_t c = index_of<MODEL>(P, bra);
			if (!col_selected(P, c)) continue;
			col[q] = (int32_t)(c - shift);
			T v;
			if (MODEL == ASM_HEISENBERG_S) { // no fermion sign; the value is tabulated per lowered digit
				const double av = P.amp[p * (P.twiceS + 1) + dj];
				if constexpr (sizeof(T) == 16) {
					v.re = av;
					v.im = 0.0;
				} else {
					v = av;
				}
			} else {
				const int par = (__popcll(ket & pr.smask_ket) + __popcll(bra & pr.smask_bra) + pr.sign_const) & 1;
				const double sg = par ? -1.0 : 1.0;
				if constexpr (sizeof(T) == 16) {
					v.re = pr.amp_re * sg;
					v.im = pr.real_only ? 0.0 : pr.amp_im * sg;
				} else {
					v = pr.amp_re * sg;
				}
			}
			val[q] = v;
			q++;
		}
	}
}


// x = beta x + alpha H y with NOTHING stored: every row re-derives its entries from the term list, as the reference's on-the-fly
// plug-in does (InternalProductOnTheFly.h:120-123 -> HubbardHelper::matrixVectorProduct, HubbardHelper.h:105-134: the same
// setHoppingTerm / setJTermOffDiagonal walk per row and call).  One thread per row; O(terms) tests and an O(L) ranking per entry.
// It serves what the structured matrix-free kernels (lpp_kron_kernels.h, lpp_pb_kernels.h) do not: terms that move both
// species (Model=SuperHubbardExtended, HubbardHelper.h:282-330).
template <int MODEL, typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void k_asm_apply(AsmParams P, const T* __restrict__ y, T* __restrict__ x, double* __restrict__ partial, EpiScale sc)
{
	__shared__ double smem[kBlock / 64];
	double alpha, beta;
	epi_coeffs(sc, alpha, beta);
	double dot = 0.0;
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const int64_t row = P.row0 + r;
		const uint64_t ket = state_of<MODEL>(P, row);
		const T yc = y[r];
		T acc = VT<T>::zero();
		{
			const double d = diag_of<MODEL>(P, ket);
			if constexpr (sizeof(T) == 16) {
				acc.re = d * yc.re;
				acc.im = d * yc.im;
			} else {
				acc = d * yc;
			}
		}
		for (int p = 0; p < P.nproc; p++) {
			const Proc& pr = P.procs[p];
			uint64_t bra;
			int dj;
			if (!proc_bra<MODEL>(P, pr, ket, bra, dj)) continue;
			const int64_t c = index_of<MODEL>(P, bra);
			const int par = (__popcll(ket & pr.smask_ket) + __popcll(bra & pr.smask_bra) + pr.sign_const) & 1;
			const double sg = par ? -1.0 : 1.0;
			T v;
			if constexpr (sizeof(T) == 16) {
				v.re = pr.amp_re * sg;
				v.im = pr.real_only ? 0.0 : pr.amp_im * sg;
			} else {
				v = pr.amp_re * sg;
			}
			VT<T>::mac(acc, v, y[c - P.row0]);
		}
		const T xv = epi_lin(beta, x[r], alpha, acc);
		x[r] = xv;
		if (DOT) dot += VT<T>::dot_re(yc, xv);
	}
	if (DOT) {
		const double s = block_sum(dot, smem);
		if (threadIdx.x == 0) partial[blockIdx.x] = s;
	}
}

// ---------------------------------------------------------------------------------------------
// Product-basis layout (lpp_pb_kernels.h): the diagonal of every row straight from the state, never through a CSR.
// Pass 1 collects the distinct diagonal values (same open-addressing table as k_dict_collect), pass 2 writes one
// dictionary code per row into the pitched code array.  diag_of adds in the reference's loop order, so the decoded
// doubles are the ones HubbardHelper::calcDiagonalElements produces (HubbardHelper.h:138-189), bit for bit.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_pb_diag_collect(AsmParams P, unsigned long long* table, int* overflow)
{
	unsigned long long last = kDictEmpty;
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const uint64_t ket = state_of<MODEL>(P, P.row0 + r);
		const unsigned long long key = (unsigned long long)__double_as_longlong(diag_of<MODEL>(P, ket));
		if (key == last) continue;
		last = key;
		unsigned h = dict_hash(key);
		int probes = 0;
		for (; probes < kDictTable; probes++) {
			const unsigned long long cur = table[h];
			if (cur == key) break;
			if (cur == kDictEmpty) {
				const unsigned long long old = atomicCAS(&table[h], kDictEmpty, key);
				if (old == kDictEmpty || old == key) break;
			}
			h = (h + 1) & (kDictTable - 1);
		}
		if (probes == kDictTable) *overflow = 1;
	}
}

template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_pb_diag_codes(AsmParams P, int64_t pitch, const double* __restrict__ dict, int ndict,
                                                           uint8_t* __restrict__ dcode, int64_t blk0 = 0, const int32_t* __restrict__ inv = nullptr, int dup = 0)
{
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const int64_t row = P.row0 + r;
		const uint64_t ket = state_of<MODEL>(P, row);
		const int64_t b = row / P.n_up, i = row - b * P.n_up;
		// blk0: first block this rank holds; inv: where the layout stores position i of a block (PbState::inv)
		const uint8_t code = (uint8_t)dict_code(dict, ndict, diag_of<MODEL>(P, ket));
		if (dup) { // complex hoppings: a position is two doubles (re, im), both carry the code
			dcode[(b - blk0) * pitch + 2 * i] = code;
			dcode[(b - blk0) * pitch + 2 * i + 1] = code;
		} else
			dcode[(b - blk0) * pitch + (inv ? inv[i] : i)] = code;
	}
}

// the diagonal as plain doubles (more than 256 distinct values), pitched like the vectors
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_pb_diag_values(AsmParams P, int64_t pitch, double* __restrict__ dval, int64_t blk0 = 0,
                                                            const int32_t* __restrict__ inv = nullptr, int dup = 0)
{
	for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < P.nloc; r += (int64_t)gridDim.x * kBlock) {
		const int64_t row = P.row0 + r;
		const uint64_t ket = state_of<MODEL>(P, row);
		const int64_t b = row / P.n_up, i = row - b * P.n_up;
		const double dv = diag_of<MODEL>(P, ket);
		if (dup) {
			dval[(b - blk0) * pitch + 2 * i] = dv;
			dval[(b - blk0) * pitch + 2 * i + 1] = dv;
		} else
			dval[(b - blk0) * pitch + (inv ? inv[i] : i)] = dv;
	}
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of int64 (three phases, chunk = 2048 elements per block)
// ---------------------------------------------------------------------------------------------
constexpr int kScanChunk = 2048;

static __global__ __launch_bounds__(kBlock) void k_scan_block_sums(const int64_t* __restrict__ in, int64_t n,
                                                                    int64_t* __restrict__ block_sums)
{
	__shared__ double smem_unused[1];
	(void)smem_unused;
	__shared__ long long sm[kBlock / 64];
	const int64_t base = (int64_t)blockIdx.x * kScanChunk;
	long long s = 0;
	for (int k = threadIdx.x; k < kScanChunk; k += kBlock)
		if (base + k < n) s += in[base + k];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t = 0;
		for (int i = 0; i < kBlock / 64; i++) t += sm[i];
		block_sums[blockIdx.x] = t;
	}
}

// single block: exclusive scan of block_sums in place; total -> *total_out
static __global__ __launch_bounds__(kBlock) void k_scan_sums(int64_t* __restrict__ block_sums, int64_t nblk,
                                                              int64_t* __restrict__ total_out)
{
	__shared__ long long sm[kBlock];
	__shared__ long long carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int64_t base = 0; base < nblk; base += kBlock) {
		const int64_t i = base + threadIdx.x;
		const long long v = (i < nblk) ? block_sums[i] : 0;
		sm[threadIdx.x] = v;
		__syncthreads();
		for (int off = 1; off < kBlock; off <<= 1) { // Hillis-Steele inclusive scan
			long long t = (threadIdx.x >= off) ? sm[threadIdx.x - off] : 0;
			__syncthreads();
			sm[threadIdx.x] += t;
			__syncthreads();
		}
		if (i < nblk) block_sums[i] = carry + sm[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 0) carry += sm[kBlock - 1];
		__syncthreads();
	}
	if (threadIdx.x == 0) *total_out = carry;
}

// per chunk: out[i] = block_offset + exclusive prefix inside the chunk (in may alias out)
static __global__ __launch_bounds__(kBlock) void k_scan_apply(const int64_t* in, int64_t n,
                                                               const int64_t* __restrict__ block_sums, int64_t* out)
{
	__shared__ long long sm[kBlock];
	constexpr int per = kScanChunk / kBlock; // 8 consecutive elements per thread
	const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * per;
	long long v[per];
	long long s = 0;
#pragma unroll
	for (int k = 0; k < per; k++) {
		v[k] = (base + k < n) ? in[base + k] : 0;
		s += v[k];
	}
	sm[threadIdx.x] = s;
	__syncthreads();
	for (int off = 1; off < kBlock; off <<= 1) {
		long long t = (threadIdx.x >= off) ? sm[threadIdx.x - off] : 0;
		__syncthreads();
		sm[threadIdx.x] += t;
		__syncthreads();
	}
	long long run = block_sums[blockIdx.x] + sm[threadIdx.x] - s;
#pragma unroll
	for (int k = 0; k < per; k++) {
		if (base + k < n) out[base + k] = run;
		run += v[k];
	}
}

} // namespace lpp
