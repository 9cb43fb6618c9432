// lpp_spmv_kernels.h -- the stored-matrix products x += H y: row-group (K1), sliced (K2) and LDS-window (K3) kernels,
// with the value dictionary, shared-offset entries, 16-bit block-local columns and the block template.
#pragma once
#include "lpp_common.h"

namespace lpp {

// ---------------------------------------------------------------------------------------------
// K1: row-group CSR SpMV   x[row] += sum_k val[k] * src[col[k]]   (+ fused partial of Re<ydot|x>)
//
// G lanes cooperate on one row (G = 4..64, chosen from nnz/row); the 64/G rows of a wave are
// consecutive, so the wave's val/col reads cover one contiguous CSR range.  Up to 4 strided
// chunks are issued per lane before the dependent gathers to keep >= 4 loads in flight.
// Row owners write x (race-free by construction, like the reference's per-row threads,
// HubbardHelper.h:119-129).  Grid-stride over rows, so consecutive blocks work on neighbouring
// rows at the same time (x-gather locality in L2 / Infinity Cache).
// ---------------------------------------------------------------------------------------------
template <typename T, int G, bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_rowgroup(int64_t nrows, const int64_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const T* __restrict__ val, const T* __restrict__ src,
                                                           T* __restrict__ x, const T* __restrict__ ydot,
                                                           double* __restrict__ partial, EpiScale sc)
{
	__shared__ double smem[kBlock / 64];
	double alpha, beta;
	epi_coeffs(sc, alpha, beta);
	const int lig = threadIdx.x % G;
	const int64_t ngroups = (int64_t)gridDim.x * (kBlock / G);
	double dot = 0.0;
	for (int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G; row < nrows; row += ngroups) {
		const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
		T acc = VT<T>::zero();
		for (int64_t p = p0 + lig; p < p1; p += 4 * G) {
			int32_t c[4];
			T v[4];
			bool ok[4];
#pragma unroll
			for (int k = 0; k < 4; k++) { // unconditional loads (clamped index), products selected afterwards
				const int64_t pk = p + (int64_t)k * G;
				ok[k] = pk < p1;
				const int64_t q = ok[k] ? pk : p;
				c[k] = col[q];
				v[k] = val[q];
			}
			T g[4];
#pragma unroll
			for (int k = 0; k < 4; k++) g[k] = src[c[k]];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				T t = VT<T>::zero();
				VT<T>::mac(t, v[k], g[k]);
				acc = VT<T>::add(acc, ok[k] ? t : VT<T>::zero());
			}
		}
#pragma unroll
		for (int off = G / 2; off > 0; off >>= 1) acc = VT<T>::add(acc, VT<T>::shfl_down(acc, off, G));
		if (lig == 0) {
			const T xv = epi_lin(beta, x[row], alpha, acc);
			x[row] = xv;
			if (DOT) dot += VT<T>::dot_re(ydot[row], xv);
		}
	}
	if (DOT) {
		const double r = block_sum(dot, smem);
		if (threadIdx.x == 0) partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// K2/K3: sliced ("wave-interleaved") CSR SpMV, optionally with an LDS-staged source window.
//
// Device-internal layout built once from the CSR (k_slice_*): rows are grouped in row blocks of B
// rows (B = N_up for the Hubbard product basis, i.e. one down-configuration; a generic power of
// two otherwise) and every block in slices of 64 rows (one wave; the last slice of a block may be
// short).  Inside a slice the entries are stored slot-major and COMPACT: all first entries of
// the rows that have one, then all second entries, ... -- the same bytes as CSR, no padding, no
// row permutation.  Lane r owns one row; at slot k the active lanes (len > k) read a dense,
// coalesced run of val/col, a lane's position in the run being the popcount of the active mask
// below it (ballot).  The gather src[col] has lanes = consecutive rows, which for product bases
// (Hubbard down-hops: col = row + const*N_up) is itself a coalesced 512-byte read.
//
// WINDOW (K3): a 1024-thread workgroup owns one row block at a time and stages the source
// entries of the block's own column range [r0, r0+B) in LDS (<= 156 KB of the CU's 160 KB);
// gathers that fall in the window (Hubbard: the diagonal and every up-hop) are served by
// ds_read_b64 instead of random 8-byte global loads, which removes their L2 misses (each one
// pulls a 128-byte line) -- measured 147 GB -> ~106 GB of fabric reads per SpMV at 4x4 Hubbard.
//
// The slot loop is software-pipelined: the col/val loads of batch b+1 are issued right after the
// gathers of batch b, so one dependent round trip per batch is exposed instead of two, and every
// load is UNCONDITIONAL (an inactive lane reads the slot's first entry and its product is
// discarded by a select): predicated loads made hipcc emit a branch and s_waitcnt vmcnt(0) per
// load, i.e. a single load in flight per wave.
// ---------------------------------------------------------------------------------------------
struct SliceGeom {
	int64_t nrows;
	int64_t B; // rows per block
	int32_t spb; // slices per block = ceil(B/64)
	int64_t nblocks;
	int64_t nslices; // nblocks * spb
};

__device__ __forceinline__ void slice_rows(const SliceGeom& g, int64_t s, int64_t& row0, int& nvalid)
{
	const int64_t blk = s / g.spb;
	const int j = (int)(s - blk * g.spb);
	row0 = blk * g.B + (int64_t)j * 64;
	const int64_t end = min((blk + 1) * g.B, g.nrows);
	const int64_t nv = end - row0;
	nvalid = nv < 0 ? 0 : (nv > 64 ? 64 : (int)nv);
}

template <int NW> __device__ __forceinline__ double block_sum_n(double v, double* smem)
{
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	__syncthreads();
	if (lane == 0) smem[w] = v;
	__syncthreads();
	double r = 0;
	if (threadIdx.x == 0) {
#pragma unroll
		for (int i = 0; i < NW; i++) r += smem[i];
	}
	return r;
}

template <typename T> struct SlicedArgs {
	SliceGeom g;
	const int64_t* slice_ptr;
	const int32_t* row_len;
	const int32_t* col;
	const T* val; // plain values (null when the value dictionary is used)
	const uint32_t* codes; // packed dictionary codes, [slice][slot group][lane]
	const int64_t* code_ptr; // first code word of each slice
	const double* dict; // <= 256 distinct doubles
	const T* src;
	T* x;
	const T* ydot;
	double* partial;
	int xcd_map;
	EpiScale sc;
	// shared-offset entries ("diagonals", see k_dia_split): slice s owns dia_off/dia_val[s*dia_stride .. +dia_stride),
	// unused places hold kDiaNone; dia_stride == 0: none
	int dia_stride;
	const int32_t* dia_off; // column - row, the same for every row of the slice
	const T* dia_val;
	// block-periodic structure (see k_tmpl_check): 1 = every row block has the row lengths and block-local columns of
	// block 0, so slice_ptr / row_len / col describe ONE block; 2 = the value codes repeat as well (codes / code_ptr
	// describe one block too; the diagonal, which does differ, travels in dcode)
	int tmpl;
	// diagonal split off the per-row entries: one dictionary code per real component and row (null: not split off)
	const uint8_t* dcode;
	// packed copy of a level-2 template (k_tmpl_pack): one 32-bit word per entry = 16-bit local column | code(s) << 16,
	// slot-major [slot][lane] per slice, every slice padded to a multiple of 8 slots with (own row, code 0 = +0.0).
	// The inner loop is then one coalesced 4-byte load, two LDS reads and an FMA per entry: no row lengths, ballots or
	// lane prefixes (measured on the compact walk: 15 VALU instructions per entry, and the 2-byte column loads cost the
	// L1 as many tag accesses as 8-byte ones).  null: walk the compact layout.
	const uint32_t* tw;
	const int32_t* tw_off; // first word of template slice j
	const int32_t* tw_len; // its padded slots
	// rows of this matrix are a reordering of the vector's rows (split-panel layout, k_split_count): x and ydot of row r live at
	// rowmap[r].  null: identity.  Only with plain per-row entries (no shared offsets, no diagonal codes).
	const int32_t* rowmap;
	// plain-format window layout with PITCHED vectors (round 5): row block b of the vectors starts at element b * (g.B + pad), so that the
	// 512-byte runs the slices gather from other blocks start on 128-byte lines; the stored columns are pitched positions.  0: contiguous
	// (only with plain per-row entries: no shared offsets, no diagonal codes, no template)
	int32_t pad;
};

constexpr int32_t kDiaNone = INT32_MIN;
constexpr int kDiaMax = 64; // shared entries per slice (one per lane of the metadata load)

// Value dictionary ("coded" layout): the Hamiltonians of this path take very few distinct values
// (+-t, J/2, U*k, ...), so when a matrix has <= 256 distinct doubles the 8-byte value of an entry is
// replaced by an 8-bit code per real component (lossless).  Codes are packed 4 slots (real) / 2 slots
// (complex, 8+8 bits) per 32-bit word, one word per lane and slot group, padded to the slice's longest
// row so a wave reads one dense 256-byte run per slot group.
template <typename T> struct CodeTraits;
template <> struct CodeTraits<double> {
	static constexpr int kBits = 8, kSlotsPerWord = 4;
	static __device__ __forceinline__ double decode(uint32_t word, int slot_in_word, const double* dict)
	{
		return dict[(word >> (8 * slot_in_word)) & 0xffu];
	}
};
template <> struct CodeTraits<cplx> {
	static constexpr int kBits = 16, kSlotsPerWord = 2;
	static __device__ __forceinline__ cplx decode(uint32_t word, int slot_in_word, const double* dict)
	{
		const uint32_t c = (word >> (16 * slot_in_word)) & 0xffffu;
		return cplx { dict[c & 0xffu], dict[c >> 8] };
	}
};

// 32-bit-offset load relative to a wave-uniform base pointer: lets hipcc use the scalar-base addressing
// form instead of building a 64-bit vector address per load
template <typename V> __device__ __forceinline__ V ld_off32(const V* base, uint32_t index)
{
	return *(const V*)((const char*)base + (size_t)(index * (uint32_t)sizeof(V)));
}

// the same with the non-temporal hint: entry streams that are read once must not push a panel of the source vector out of L2
// (split-panel layout: 42 MB of columns and values pass while one 1.6 MB panel is being gathered from)
template <typename V> __device__ __forceinline__ V ld_off32_nt(const V* base, uint32_t index)
{
	if constexpr (sizeof(V) == 16) {
		const double* p = (const double*)((const char*)base + (size_t)(index * 16u));
		V r;
		double* q = (double*)&r;
		q[0] = __builtin_nontemporal_load(p);
		q[1] = __builtin_nontemporal_load(p + 1);
		return r;
	} else {
		return __builtin_nontemporal_load((const V*)((const char*)base + (size_t)(index * (uint32_t)sizeof(V))));
	}
}

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m)
{ // number of set bits of m below this lane (v_mbcnt_lo + v_mbcnt_hi)
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Accumulate one slice with one wave: returns sum_k val_k * src[col_k] of this lane's row.
// (len, base, cbase) = this lane's row length, the slice's first entry and first code word, prefetched by
// the caller; `safe` is a valid source index used by lanes whose gather is served from the LDS window.
// Gathers use 32-bit byte offsets (the engine only selects these kernels for vectors < 4 GiB).
// Per-entry instruction count matters here (measured: ~31 VALU instructions per entry made the kernel
// issue-bound for 25-50 % of its time): mbcnt for the lane prefix, scalar base pointers, wave-uniform
// skips of the global gather when a whole slot is inside the window (and of the LDS read when none is),
// and -- coded layout -- no select at all: inactive lanes decode code 0 == +0.0.
// LOCAL16 (window kernel only): every entry's column lies in the row block's own window and columns are stored as
// 16-bit window-local indices -- 2 bytes per entry and no in-window test, select or global gather in the loop.
template <typename T, bool WINDOW, bool CODED, int U, bool LOCAL16 = false>
__device__ __forceinline__ T sliced_accumulate(const SlicedArgs<T>& a, int len, int64_t base, int64_t cbase, const T* lds,
                                               int32_t r0, uint32_t wlen, const double* dict, int32_t safe)
{
	constexpr int SPW = CodeTraits<T>::kSlotsPerWord;
	constexpr int NW = (U + SPW - 1) / SPW; // code words per batch
	const int lane = threadIdx.x & 63;
	int maxlen = len;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
	const int nb = (maxlen + U - 1) / U;
	// base / cbase are the same in every lane: move them to SGPRs so that every stream address is
	// (scalar base pointer + 32-bit lane offset) instead of 64-bit vector arithmetic per load
	const int64_t base_u = ((int64_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
	const int64_t cbase_u = ((int64_t)__builtin_amdgcn_readfirstlane((int)(cbase >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)cbase);
	const int32_t* colp = LOCAL16 ? nullptr : a.col + base_u;
	const uint16_t* colp16 = LOCAL16 ? (const uint16_t*)a.col + base_u : nullptr;
	const T* valp = CODED ? nullptr : a.val + base_u;
	const uint32_t* codep = CODED ? a.codes + cbase_u : nullptr;
	uint32_t run = 0; // entries of this slice consumed so far (wave-uniform)
	const bool nt_ = !WINDOW && a.rowmap != nullptr; // the panel-major part of a split-panel matrix: streams read with the non-temporal hint
	T acc = VT<T>::zero();
	int32_t c0[U], c1[U];
	T v0[CODED ? 1 : U], v1[CODED ? 1 : U];
	uint32_t w0[CODED ? NW : 1], w1[CODED ? NW : 1];
#define LPP_LOAD_BATCH(K0, C, V, W)                                                                                   \
	_Pragma("unroll") for (int u = 0; u < U; u++)                                                                     \
	{                                                                                                                 \
		const bool on_ = len > (K0) + u;                                                                              \
		const unsigned long long m_ = __ballot(on_);                                                                  \
		const uint32_t p_ = run + (on_ ? lane_prefix(m_) : 0u);                                                       \
		C[u] = LOCAL16 ? (int32_t)ld_off32(colp16, p_) : (nt_ ? ld_off32_nt(colp, p_) : ld_off32(colp, p_));          \
		/* lanes past their row's end read the window's first element instead of whatever column the stream holds there: the slots   */ \
		/* behind a slice's longest row (up to U - 1 per slice) gathered a line of the source vector each, 2.3 GB per product at config 2 */ \
		if (WINDOW && !LOCAL16) C[u] = on_ ? C[u] : r0;                                                                 \
		if (!CODED) {                                                                                                 \
			const T t_ = nt_ ? ld_off32_nt(valp, p_) : ld_off32(valp, p_);                                            \
			V[u] = on_ ? t_ : VT<T>::zero();                                                                          \
		}                                                                                                             \
		run += (uint32_t)__popcll(m_);                                                                                \
	}                                                                                                                 \
	if (CODED) {                                                                                                      \
		_Pragma("unroll") for (int q = 0; q < NW; q++) W[q] = ld_off32(codep, (uint32_t)(((K0) / SPW + q) << 6) + (uint32_t)lane); \
	}
	if (nb > 0) { LPP_LOAD_BATCH(0, c0, v0, w0) }
	for (int b = 0; b < nb; b++) {
		T g[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			if (LOCAL16) {
				g[u] = lds[c0[u]];
			} else if (WINDOW) {
				const uint32_t d = (uint32_t)(c0[u] - r0);
				const bool inw = d < wlen;
				const unsigned long long min_ = __ballot(inw);
				T gl = VT<T>::zero(), gg = VT<T>::zero();
				if (min_ != 0ull) gl = lds[inw ? d : (uint32_t)lane]; // wave-uniform branches
				if (min_ != ~0ull) gg = ld_off32(a.src, (uint32_t)(inw ? safe : c0[u]));
				g[u] = inw ? gl : gg;
			} else {
				g[u] = ld_off32(a.src, (uint32_t)c0[u]);
			}
		}
		if (b + 1 < nb) { LPP_LOAD_BATCH((b + 1) * U, c1, v1, w1) }
#pragma unroll
		for (int u = 0; u < U; u++) {
			// inactive lanes carry a zero value (plain: selected at load; coded: code 0 decodes to +0.0)
			const T vv = CODED ? CodeTraits<T>::decode(w0[u / SPW], u % SPW, dict) : v0[u];
			VT<T>::mac(acc, vv, g[u]);
		}
#pragma unroll
		for (int u = 0; u < U; u++) c0[u] = c1[u];
		if (CODED) {
#pragma unroll
			for (int q = 0; q < NW; q++) w0[q] = w1[q];
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) v0[u] = v1[u];
		}
	}
#undef LPP_LOAD_BATCH
	return acc;
}

// Shared-offset entries of a slice: entry d contributes val_d * src[row + off_d] to EVERY row of the slice, so the gather
// is one contiguous 64-element run and needs no column load.  The slice's (off, val) list is fetched by ONE vector load
// (lane l takes entry l, a slice ahead, together with the other slice metadata) and handed out with v_readlane, i.e.
// offsets and values are scalar operands.  The first 8*kChunks gathers are requested before the slice's per-row
// entries are walked and consumed after them.
template <typename T> struct DiaMeta {
	int32_t off; // lane l: offset of shared entry l, kDiaNone past the end
	T val;
};

__device__ __forceinline__ double readlane_t(double v, int l)
{
	const long long b = __double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ cplx readlane_t(cplx v, int l) { return cplx { readlane_t(v.re, l), readlane_t(v.im, l) }; }

template <typename T> __device__ __forceinline__ void dia_meta(const SlicedArgs<T>& a, int64_t s, DiaMeta<T>& m)
{
	m.off = kDiaNone;
	m.val = VT<T>::zero();
	if (a.dia_stride > 0) { // wave-uniform
		const int lane = threadIdx.x & 63;
		const int64_t i = s * a.dia_stride + min(lane, a.dia_stride - 1);
		const int32_t o = a.dia_off[i];
		m.val = a.dia_val[i];
		m.off = lane < a.dia_stride ? o : kDiaNone;
	}
}

template <typename T> struct DiaPre {
	static constexpr int kChunks = sizeof(T) == 8 ? 2 : 1; // chunks of 8 gathers kept in flight
	T g[8 * kChunks];
};

template <typename T>
__device__ __forceinline__ void dia_request(const SlicedArgs<T>& a, const DiaMeta<T>& m, int dcnt, uint32_t row, DiaPre<T>& pre)
{
#pragma unroll
	for (int ch = 0; ch < DiaPre<T>::kChunks; ch++) {
		if (ch * 8 < dcnt) { // wave-uniform
#pragma unroll
			for (int q = 0; q < 8; q++) {
				// clamped: the tail of a chunk repeats the last entry (its value is selected to 0 in dia_consume)
				const int32_t o = __builtin_amdgcn_readlane(m.off, min(ch * 8 + q, dcnt - 1));
				pre.g[ch * 8 + q] = ld_off32(a.src, row + (uint32_t)o);
			}
		}
	}
}

template <typename T>
__device__ __forceinline__ void dia_consume(const SlicedArgs<T>& a, const DiaMeta<T>& m, int dcnt, uint32_t row, const DiaPre<T>& pre, T& acc)
{
	constexpr int NPRE = 8 * DiaPre<T>::kChunks;
	for (int d0 = NPRE; d0 < dcnt; d0 += 8) { // rare: more shared entries than prefetch places
		T g[8];
#pragma unroll
		for (int q = 0; q < 8; q++) g[q] = ld_off32(a.src, row + (uint32_t)__builtin_amdgcn_readlane(m.off, min(d0 + q, dcnt - 1)));
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const T v = d0 + q < dcnt ? readlane_t(m.val, min(d0 + q, dcnt - 1)) : VT<T>::zero();
			VT<T>::mac(acc, v, g[q]);
		}
	}
#pragma unroll
	for (int ch = 0; ch < DiaPre<T>::kChunks; ch++) {
		if (ch * 8 < dcnt) {
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const int d = ch * 8 + q;
				const T v = d < dcnt ? readlane_t(m.val, min(d, dcnt - 1)) : VT<T>::zero();
				VT<T>::mac(acc, v, pre.g[d]);
			}
		}
	}
}

// x += sum over the packed template entries of slice j (window kernel, see SlicedArgs::tw)
template <typename T>
__device__ __forceinline__ T tmpl_accumulate(const SlicedArgs<T>& a, int j, const T* lds, const double* dict)
{
	const int lane = threadIdx.x & 63;
	const uint32_t* wp = a.tw + a.tw_off[j] + lane;
	const int ml = a.tw_len[j]; // multiple of 8
	T acc = VT<T>::zero();
	uint32_t w0[8], w1[8];
	if (ml > 0) {
#pragma unroll
		for (int q = 0; q < 8; q++) w0[q] = wp[q * 64];
	}
	for (int k = 0; k < ml; k += 8) {
		if (k + 8 < ml) {
#pragma unroll
			for (int q = 0; q < 8; q++) w1[q] = wp[(k + 8 + q) * 64];
		}
#pragma unroll
		for (int q = 0; q < 8; q++) VT<T>::mac(acc, CodeTraits<T>::decode(w0[q] >> 16, 0, dict), lds[w0[q] & 0xffffu]);
#pragma unroll
		for (int q = 0; q < 8; q++) w0[q] = w1[q];
	}
	return acc;
}

// process one slice with one wave (x[row] += acc); returns this lane's contribution to Re<ydot|x>.
template <typename T, bool DOT, bool WINDOW, bool CODED, int U, bool LOCAL16 = false>
__device__ __forceinline__ double sliced_one(const SlicedArgs<T>& a, int64_t row0, int nvalid, int len, int64_t base, int64_t cbase,
                                             const T* lds, int32_t r0, uint32_t wlen, const double* dict, double alpha, double beta,
                                             const DiaMeta<T>& dm, int64_t rowshift = 0)
{
	if (nvalid == 0) return 0.0; // wave-uniform
	const int lane = threadIdx.x & 63;
	const bool valid = lane < nvalid;
	// rowshift (pitched vectors, SlicedArgs::pad): `row` counts positions of the vectors from here on; the slice's metadata was read by the caller
	const int64_t row = row0 + rowshift + (valid ? lane : 0);
	const int64_t xrow = a.rowmap ? (int64_t)a.rowmap[row] : row; // where this row's x and y live
	// the row's old x and y are requested first: they are the oldest loads in flight and have landed
	// long before the epilogue needs them
	const T xold = a.x[xrow];
	T yv = VT<T>::zero();
	if (DOT) yv = a.ydot[xrow]; // (taking it from the LDS window when ydot is the source vector measured slower: 18.1-18.9 against 16.9-18.4 ms on the plain-format leg)
	uint32_t dc = 0; // code(s) of the diagonal value, when it travels apart from the per-row entries
	if (CODED && a.dcode) dc = sizeof(T) == 16 ? (uint32_t)((const uint16_t*)a.dcode)[row] : (uint32_t)a.dcode[row];
	DiaPre<T> pre;
	// shared entries of the slice: a leading run of places (global gathers) and a trailing run (inside the LDS window)
	const unsigned long long dmask = __ballot(dm.off != kDiaNone);
	const int dcnt = dmask == ~0ull ? 64 : __ffsll((long long)~dmask) - 1; // wave-uniform
	int wcnt = 0;
	if (WINDOW && a.dia_stride > 0) wcnt = __clzll((long long)~(dmask << (64 - a.dia_stride)));
	dia_request<T>(a, dm, dcnt, (uint32_t)row, pre);
	T acc;
	if (LOCAL16 && CODED && a.tw) // wave-uniform
		acc = tmpl_accumulate<T>(a, (int)((row0 - r0) >> 6), lds, dict);
	else
		acc = sliced_accumulate<T, WINDOW, CODED, U, LOCAL16>(a, len, base, cbase, lds, r0, wlen, dict, (int32_t)row);
	dia_consume<T>(a, dm, dcnt, (uint32_t)row, pre, acc);
	if (WINDOW) {
		for (int i = 0; i < wcnt; i++) { // contiguous 64-element runs of the window: conflict-free LDS reads, scalar offset/value
			const int place = a.dia_stride - 1 - i;
			const int32_t o = __builtin_amdgcn_readlane(dm.off, place);
			VT<T>::mac(acc, readlane_t(dm.val, place), lds[(uint32_t)((int32_t)row - r0 + o)]);
		}
	}
	if (CODED && a.dcode) {
		const T ys = WINDOW ? lds[(uint32_t)((int32_t)row - r0)] : ld_off32(a.src, (uint32_t)row);
		VT<T>::mac(acc, CodeTraits<T>::decode(dc, 0, dict), ys);
	}
	double d = 0.0;
	if (valid) {
		const T xv = epi_lin(beta, xold, alpha, acc);
		a.x[xrow] = xv;
		if (DOT) d = VT<T>::dot_re(yv, xv);
	}
	return d;
}

// metadata of slice s for this lane (row_len is read unconditionally from a clamped row)
template <typename T, bool CODED>
__device__ __forceinline__ void slice_meta(const SlicedArgs<T>& a, int64_t s, int64_t& row0, int& nvalid, int& len, int64_t& base,
                                           int64_t& cbase)
{
	const int lane = threadIdx.x & 63;
	slice_rows(a.g, s, row0, nvalid);
	len = 0;
	base = 0;
	cbase = 0;
	if (CODED && a.tw) return; // the packed template is walked instead (tmpl_accumulate): no per-row metadata needed
	const int64_t r = (lane < nvalid) ? row0 + lane : min(row0, a.g.nrows - 1);
	// block-periodic structure: lengths and column stream of the same slice of block 0 (L2-resident)
	const int64_t blk = a.tmpl ? s / a.g.spb : 0;
	const int l = a.row_len[r - blk * a.g.B];
	len = (lane < nvalid) ? l : 0;
	base = a.slice_ptr[s - blk * a.g.spb];
	cbase = CODED ? a.code_ptr[a.tmpl == 2 ? s - blk * a.g.spb : s] : 0;
}

// the dictionary lives in LDS (2 KB); decode reads are mostly broadcasts (few distinct values)
template <bool CODED> __device__ __forceinline__ void load_dict(double* dict_s, const double* dict)
{
	if (CODED) {
		for (int i = threadIdx.x; i < 256; i += blockDim.x) dict_s[i] = dict[i];
		__syncthreads();
	}
}

// K2: no window, 256-thread blocks, waves walk slices (grid-stride, or one contiguous eighth of
// the slices per XCD: blocks b and b+8 share an XCD under round-robin dispatch -- speed only).
template <typename T, bool DOT, bool CODED, int U>
__global__ __launch_bounds__(kBlock) void k_spmv_sliced(SlicedArgs<T> a)
{
	__shared__ double smem[kBlock / 64];
	__shared__ double dict_s[CODED ? 256 : 1];
	load_dict<CODED>(dict_s, a.dict);
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	int64_t s_begin, s_end, s_stride;
	if (a.xcd_map && (gridDim.x & 7) == 0) {
		const int64_t chunk = (a.g.nslices + 7) / 8;
		const int xcd = blockIdx.x & 7;
		s_begin = xcd * chunk + (int64_t)(blockIdx.x >> 3) * (kBlock / 64) + (threadIdx.x >> 6);
		s_end = min((int64_t)(xcd + 1) * chunk, a.g.nslices);
		s_stride = (int64_t)(gridDim.x >> 3) * (kBlock / 64);
	} else {
		s_begin = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
		s_end = a.g.nslices;
		s_stride = (int64_t)gridDim.x * (kBlock / 64);
	}
	double dot = 0.0;
	int64_t row0 = 0, base = 0, cbase = 0;
	int nvalid = 0, len = 0;
	DiaMeta<T> dm { kDiaNone, VT<T>::zero() };
	if (s_begin < s_end) {
		slice_meta<T, CODED>(a, s_begin, row0, nvalid, len, base, cbase);
		dia_meta<T>(a, s_begin, dm);
	}
	for (int64_t s = s_begin; s < s_end; s += s_stride) {
		// prefetch the next slice's metadata before working on this one
		int64_t row0n = 0, basen = 0, cbasen = 0;
		int nvalidn = 0, lenn = 0;
		DiaMeta<T> dmn { kDiaNone, VT<T>::zero() };
		if (s + s_stride < s_end) {
			slice_meta<T, CODED>(a, s + s_stride, row0n, nvalidn, lenn, basen, cbasen);
			dia_meta<T>(a, s + s_stride, dmn);
		}
		dot += sliced_one<T, DOT, false, CODED, U>(a, row0, nvalid, len, base, cbase, nullptr, 0, 0, dict_s, alpha, beta, dm);
		row0 = row0n;
		base = basen;
		cbase = cbasen;
		nvalid = nvalidn;
		len = lenn;
		dm = dmn;
	}
	if (DOT) {
		const double r = block_sum(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// one lane takes the next slice index from the workgroup's LDS counter and broadcasts it to its wave
__device__ __forceinline__ int next_slice_claim(int* counter)
{
	int v = 0;
	if ((threadIdx.x & 63) == 0) v = atomicAdd(counter, 1);
	return __builtin_amdgcn_readfirstlane(v);
}

// K3: LDS window.  One 1024-thread workgroup per CU walks row blocks; dynamic LDS = B elements.
constexpr int kWinThreads = 1024;
template <typename T, bool DOT, bool CODED, int U, bool LOCAL16>
__global__ __launch_bounds__(kWinThreads) void k_spmv_window(SlicedArgs<T> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	T* lds = (T*)lds_raw;
	__shared__ double smem[kWinThreads / 64];
	__shared__ double dict_s[CODED ? 256 : 1];
	__shared__ int next_slice;
	load_dict<CODED>(dict_s, a.dict);
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	int64_t b_begin, b_end, b_stride;
	if (a.xcd_map && (gridDim.x & 7) == 0) {
		const int64_t chunk = (a.g.nblocks + 7) / 8;
		const int xcd = blockIdx.x & 7;
		b_begin = xcd * chunk + (blockIdx.x >> 3);
		b_end = min((int64_t)(xcd + 1) * chunk, a.g.nblocks);
		b_stride = gridDim.x >> 3;
	} else {
		b_begin = blockIdx.x;
		b_end = a.g.nblocks;
		b_stride = gridDim.x;
	}
	double dot = 0.0;
	for (int64_t blk = b_begin; blk < b_end; blk += b_stride) {
		const int64_t rshift = blk * (int64_t)a.pad; // pitched vectors: this block's rows sit rshift positions further on
		const int64_t r0 = blk * a.g.B + rshift;
		const int64_t wl = min(a.g.B, a.g.nrows - blk * a.g.B);
		__syncthreads(); // everyone is done reading the previous window
		if (threadIdx.x == 0) next_slice = 0;
		// stage the window: 8 independent loads per thread in flight (a load-wait-store loop exposed
		// one HBM round trip per element and kept all 16 waves idle for ~25 us per block)
		for (int64_t i0 = threadIdx.x; i0 < wl; i0 += 8 * kWinThreads) {
			T t[8];
#pragma unroll
			for (int q = 0; q < 8; q++) t[q] = a.src[r0 + min(i0 + (int64_t)q * kWinThreads, wl - 1)];
#pragma unroll
			for (int q = 0; q < 8; q++)
				if (i0 + (int64_t)q * kWinThreads < wl) lds[i0 + (int64_t)q * kWinThreads] = t[q];
		}
		__syncthreads();
		// slices are handed out dynamically (LDS counter): with 201 slices on 16 waves a static split leaves
		// the waves that got 12 instead of 13 slices idle at the block's closing barrier (~8 % of the time)
		int64_t row0 = 0, base = 0, cbase = 0;
		int nvalid = 0, len = 0;
		DiaMeta<T> dm { kDiaNone, VT<T>::zero() };
		int j = next_slice_claim(&next_slice);
		if (j < a.g.spb) {
			slice_meta<T, CODED>(a, blk * a.g.spb + j, row0, nvalid, len, base, cbase);
			dia_meta<T>(a, blk * a.g.spb + j, dm);
		}
		while (j < a.g.spb) {
			const int jn = next_slice_claim(&next_slice);
			int64_t row0n = 0, basen = 0, cbasen = 0;
			int nvalidn = 0, lenn = 0;
			DiaMeta<T> dmn { kDiaNone, VT<T>::zero() };
			if (jn < a.g.spb) {
				slice_meta<T, CODED>(a, blk * a.g.spb + jn, row0n, nvalidn, lenn, basen, cbasen);
				dia_meta<T>(a, blk * a.g.spb + jn, dmn);
			}
			dot += sliced_one<T, DOT, true, CODED, U, LOCAL16>(a, row0, nvalid, len, base, cbase, lds, (int32_t)r0, (uint32_t)wl, dict_s, alpha, beta, dm, rshift);
			row0 = row0n;
			base = basen;
			cbase = cbasen;
			nvalid = nvalidn;
			len = lenn;
			dm = dmn;
			j = jn;
		}
	}
	if (DOT) {
		const double r = block_sum_n<kWinThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

} // namespace lpp
