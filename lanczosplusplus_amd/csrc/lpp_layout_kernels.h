// lpp_layout_kernels.h -- one-off kernels that turn a plain CSR into the layouts of lpp_spmv_kernels.h and back
// (lpp_engine_get_csr): slicing, shared-offset split and merge, value dictionary and codes, block template.
#pragma once
#include "lpp_spmv_kernels.h"

namespace lpp {

// CSR -> sliced layout.  Slices cover consecutive row ranges, so a slice's entries are the CSR range
// rowptr[row0] .. rowptr[row0+nvalid).  words[s] (optional) = code words of slice s = 64*ceil(maxlen/spw).
static __global__ void k_slice_meta(SliceGeom g, const int64_t* __restrict__ rowptr, int64_t* __restrict__ slice_ptr,
                                    int32_t* __restrict__ row_len, int64_t* __restrict__ words, int spw)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < g.nrows) row_len[i] = (int32_t)(rowptr[i + 1] - rowptr[i]);
	if (i < g.nslices) {
		int64_t row0;
		int nvalid;
		slice_rows(g, i, row0, nvalid);
		slice_ptr[i] = rowptr[nvalid > 0 ? row0 : g.nrows];
		if (words) {
			int64_t mx = 0;
			for (int r = 0; r < nvalid; r++) mx = max(mx, rowptr[row0 + r + 1] - rowptr[row0 + r]);
			words[i] = 64 * ((mx + 7) / 8) * (8 / spw); // padded to whole batches of 8 slots: trailing codes are 0
		}
	}
	if (i == g.nslices) {
		slice_ptr[i] = rowptr[g.nrows];
		if (words) words[i] = 0;
	}
}

// one wave per slice: scatter CSR entries into slot-major compact order (INVERSE: back to CSR order)
// L16: the sliced side holds 16-bit window-local columns (column - first row of the row block)
// tmpl (INVERSE only): the sliced column stream holds block 0 only (block-periodic structure)
// outs_first (plain format, window kernel, rows sorted by column; round 5): a row's entries that LEAVE its row block are stored in its
// first slots -- those below the block, then those above it --, the entries inside the block behind them.  In a product basis all rows of a
// block leave it the same way (the same down-hops), but in column order the entries above the block sit behind a row's in-block entries,
// whose number differs from row to row: a slot of the slice then gathers pieces of several 512-byte runs, each run is asked for by up to 7
// slots ~1 us apart, and the matrix stream renews an XCD's L2 every ~6 us -- 37.7 GB of fabric reads for 22.6 GB of gathered elements at
// BASELINE config 2 (timing-only build without the gathers: 75.3 instead of 113.0 GB).  With the leaving entries first every run is asked
// for by ONE load.  Same 12 bytes per entry; the CSR order is restored from the column values (below / inside / above the block).
template <typename T, bool INVERSE, bool L16 = false>
__global__ __launch_bounds__(kBlock) void k_slice_fill(SliceGeom g, const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col_in,
                                                        const T* __restrict__ val_in, int32_t* __restrict__ col_out,
                                                        T* __restrict__ val_out, int tmpl = 0, int outs_first = 0, int pad = 0)
{
	// pad (with outs_first only): the sliced side holds PITCHED columns -- column c of row block c / B sits at c + (c / B) * pad
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	for (int64_t s = wave0; s < g.nslices; s += nwaves) {
		int64_t row0;
		int nvalid;
		slice_rows(g, s, row0, nvalid);
		int64_t p0 = 0;
		int len = 0;
		if (lane < nvalid) {
			p0 = rowptr[row0 + lane];
			len = (int)(rowptr[row0 + lane + 1] - p0);
		}
		int64_t base = rowptr[nvalid > 0 ? row0 : g.nrows];
		// position of this slice's columns in the sliced stream (block 0's copy when the structure is block-periodic)
		int64_t cshift = 0;
		if (INVERSE && tmpl && nvalid > 0) cshift = rowptr[row0 - (s / g.spb) * g.B] - base;
		int maxlen = len;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
		// outs_first: entries of this row below / above its row block (rows sorted by column: the CSR order is below, inside, above)
		int nlo = 0, nhi = 0;
		if (outs_first && !L16) { // wave-uniform
			const int64_t w0 = (s / g.spb) * (g.B + (INVERSE ? pad : 0)), w1 = w0 + g.B; // (the sliced side's columns are pitched)
			if (!INVERSE) {
				for (int k = 0; k < len; k++) {
					const int64_t c = col_in[p0 + k];
					nlo += c < w0 ? 1 : 0;
					nhi += c >= w1 ? 1 : 0;
				}
			} else {
				int64_t b2 = base;
				for (int k = 0; k < maxlen; k++) {
					const bool on = len > k;
					const unsigned long long m = __ballot(on);
					if (on) {
						const int64_t c = col_in[b2 + __popcll(m & ((1ull << lane) - 1ull))];
						nlo += c < w0 ? 1 : 0;
						nhi += c >= w1 ? 1 : 0;
					}
					b2 += __popcll(m);
				}
			}
		}
		const int nin = len - nlo - nhi;
		for (int k = 0; k < maxlen; k++) {
			const bool on = len > k;
			const unsigned long long m = __ballot(on);
			const int pos = __popcll(m & ((1ull << lane) - 1ull));
			if (on) {
				const int32_t r0 = L16 ? (int32_t)((s / g.spb) * g.B) : 0;
				// slot k of the row holds CSR entry ek (stored order with outs_first: below the block, above it, inside it)
				const int ek = (!outs_first || L16 || k < nlo) ? k : (k < nlo + nhi ? nlo + nin + (k - nlo) : nlo + (k - nlo - nhi));
				if (INVERSE) {
					int32_t c = L16 ? (int32_t)((const uint16_t*)col_in)[base + cshift + pos] + r0 : col_in[base + cshift + pos];
					if (pad && !L16) c -= (int32_t)(c / (g.B + pad)) * pad;
					col_out[p0 + ek] = c;
					if (val_out) val_out[p0 + ek] = val_in[base + pos];
				} else {
					if (L16)
						((uint16_t*)col_out)[base + pos] = (uint16_t)(col_in[p0 + ek] - r0);
					else
						col_out[base + pos] = pad ? col_in[p0 + ek] + (int32_t)(col_in[p0 + ek] / g.B) * pad : col_in[p0 + ek];
					if (val_out) val_out[base + pos] = val_in[p0 + ek];
				}
			}
			base += __popcll(m);
		}
	}
}

// ---- shared-offset ("diagonal") entries -----------------------------------------------------------
// Product-basis Hamiltonians repeat themselves: in the Hubbard basis every row of one down-configuration block has
// the same down-hops, i.e. the entries (column - row, value) are identical for all 64 rows of a slice.  Such an entry
// is stored once per slice (12 or 20 bytes) instead of once per row, its gather needs no column load at all, and
// offset and value live in scalar registers.  The split is structural (no model knowledge) and lossless:
//   CSR = per-row "rest" entries (sliced layout as before) + per-slice shared entries, merged back by k_dia_merge.
// An entry of the slice's first row is shared when every other valid row holds an entry with the same offset and
// bit-identical value.  Rows must be strictly sorted by column (checked by k_rows_sorted; otherwise disabled).
template <typename T> __device__ __forceinline__ bool same_bits(const T& x, const T& y);
template <> __device__ __forceinline__ bool same_bits<double>(const double& x, const double& y)
{
	return __double_as_longlong(x) == __double_as_longlong(y);
}
template <> __device__ __forceinline__ bool same_bits<cplx>(const cplx& x, const cplx& y)
{
	return __double_as_longlong(x.re) == __double_as_longlong(y.re) && __double_as_longlong(x.im) == __double_as_longlong(y.im);
}

// device-side counterpart of the host CSR validation (lpp_engine_set_csr_device)
static __global__ void k_check_csr(int64_t nrows, int64_t ncols, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int* bad)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r == 0 && rowptr[0] != 0) *bad = 1;
	if (r >= nrows) return;
	const int64_t p0 = rowptr[r], p1 = rowptr[r + 1];
	if (p1 < p0) {
		*bad = 1;
		return;
	}
	bool b = false;
	for (int64_t p = p0; p < p1; p++) b |= col[p] < 0 || (int64_t)col[p] >= ncols;
	if (b) *bad = 1;
}

static __global__ void k_rows_sorted(int64_t nrows, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int* unsorted)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nrows) return;
	bool bad = false;
	for (int64_t p = rowptr[r] + 1; p < rowptr[r + 1]; p++) bad |= col[p] <= col[p - 1];
	if (bad) *unsorted = 1;
}

__device__ __forceinline__ uint32_t dict_code(const double* dict, int ndict, double v);

// One wave per slice.  FILL == false: rest_len[row] = entries the row keeps; stats[0] = max shared entries of any slice,
// stats[1] = shared entries summed over slices.  FILL == true (after the scan of rest_len): writes the rest CSR
// (rcol/rval at rrowptr) and the shared lists at dia_off/dia_val[s*stride ..] (pre-filled with kDiaNone / 0).
// win != 0: the matrix is built for the LDS-window kernel; shared entries whose whole 64-row run lies inside the row block
// are listed from the END of the slice's places (stride-1 downwards) and are read from the LDS window, the others from
// place 0 upwards and are gathered from global memory; at least one empty place separates the two groups.
// xdiag != 0: the diagonal entry of every row is taken out of the per-row entries as well (FILL: its dictionary code(s)
// go to dcode[row]); stats[2] counts rows WITHOUT a diagonal entry (the caller then repeats the count with xdiag = 0).
template <typename T, bool FILL>
__global__ __launch_bounds__(kBlock) void k_dia_split(SliceGeom g, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const T* __restrict__ val, int win, int stride, int64_t* __restrict__ rest_len,
                                                       unsigned long long* __restrict__ stats, const int64_t* __restrict__ rrowptr,
                                                       int32_t* __restrict__ rcol, T* __restrict__ rval, int32_t* __restrict__ dia_off,
                                                       T* __restrict__ dia_val, int xdiag, const double* __restrict__ dict, int ndict,
                                                       uint8_t* __restrict__ dcode)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	unsigned long long local_max = 0, local_sum = 0, local_nodiag = 0;
	// The slice's CSR range (64 consecutive rows = one contiguous run of entries) is staged in LDS first: the scan below
	// walks every row entry by entry, and from global memory that is ~70 dependent, 64-way scattered loads per slice
	// (measured: 0.27 s + 0.40 s for the two passes over the 5.8e9-entry matrix).  Slices longer than the stage keep
	// reading global memory.
	constexpr int kStage = (int)(33 * 1024 / (sizeof(int32_t) + sizeof(T)));
	__shared__ __attribute__((aligned(16))) unsigned char stage_raw[(kBlock / 64) * kStage * (sizeof(int32_t) + sizeof(T))];
	T* sval = (T*)stage_raw + (size_t)(threadIdx.x >> 6) * kStage;
	int32_t* scol = (int32_t*)((T*)stage_raw + (size_t)(kBlock / 64) * kStage) + (size_t)(threadIdx.x >> 6) * kStage;
#define LPP_COL(Q) (staged ? scol[(Q) - pfirst] : col[Q])
#define LPP_VAL(Q) (staged ? sval[(Q) - pfirst] : val[Q])
	// a per-row entry that stays: emitted to the rest CSR, or -- the diagonal, when it is split off -- to dcode
#define LPP_KEEP_ENTRY(Q)                                                                                             \
	do {                                                                                                              \
		if (xdiag && (int64_t)LPP_COL(Q) == row) {                                                                    \
			ndg++;                                                                                                    \
			if (FILL) {                                                                                               \
				const T tv_ = LPP_VAL(Q);                                                                             \
				const double* pv_ = (const double*)&tv_;                                                              \
				if (sizeof(T) == 16) {                                                                                \
					dcode[2 * row] = (uint8_t)dict_code(dict, ndict, pv_[0]);                                         \
					dcode[2 * row + 1] = (uint8_t)dict_code(dict, ndict, pv_[1]);                                     \
				} else {                                                                                              \
					dcode[row] = (uint8_t)dict_code(dict, ndict, pv_[0]);                                             \
				}                                                                                                     \
			}                                                                                                         \
		} else if (FILL) {                                                                                            \
			rcol[wp] = LPP_COL(Q);                                                                                    \
			rval[wp] = LPP_VAL(Q);                                                                                    \
			wp++;                                                                                                     \
		}                                                                                                             \
	} while (0)
	for (int64_t s = wave0; s < g.nslices; s += nwaves) {
		int64_t row0;
		int nvalid;
		slice_rows(g, s, row0, nvalid);
		if (nvalid == 0) continue;
		const bool valid = lane < nvalid;
		const int64_t row = row0 + (valid ? lane : 0);
		int ndg = 0; // diagonal entries of this row that were split off (0 or 1)
		const int64_t pbeg = rowptr[row];
		int64_t q = valid ? pbeg : 0;
		const int64_t end = valid ? rowptr[row + 1] : 0;
		const int64_t p00 = rowptr[row0];
		const int len0 = (int)(rowptr[row0 + 1] - p00);
		const int64_t pfirst = p00, plast = rowptr[row0 + nvalid];
		const bool staged = plast - pfirst <= kStage; // wave-uniform
		if (staged) {
			for (int64_t i = pfirst + lane; i < plast; i += 64) {
				scol[i - pfirst] = col[i];
				sval[i - pfirst] = val[i];
			}
		}
		const int64_t blk0 = (s / g.spb) * g.B, blk1 = blk0 + g.B;
		const unsigned long long vmask = __ballot(valid);
		int64_t wp = (FILL && valid) ? rrowptr[row] : 0;
		int nd = 0; // shared entries gathered from global memory: places 0, 1, ... of the slice's list
		int nw = 0; // shared entries whose whole run lies inside the LDS window: places stride-1, stride-2, ...
		for (int k = 0; k < len0; k++) {
			const int32_t c0 = LPP_COL(p00 + k);
			const T v0 = LPP_VAL(p00 + k);
			const int64_t off = (int64_t)c0 - row0;
			const int64_t target = row + off;
			while (q < end && (int64_t)LPP_COL(q) < target) { // entries passed over stay with the row
				LPP_KEEP_ENTRY(q);
				q++;
			}
			bool ok = valid && q < end && (int64_t)LPP_COL(q) == target;
			if (ok) ok = same_bits<T>(LPP_VAL(q), v0);
			const bool in_block = win && row0 + off >= blk0 && row0 + (nvalid - 1) + off < blk1;
			// shared by every valid row of the slice (the diagonal, when it is split off, has its own stream); one place
			// of the list always stays empty between the two groups
			if (__ballot(ok) == vmask && !(xdiag && off == 0) && nd + nw < kDiaMax - 2) {
				if (FILL && lane == 0) {
					const int64_t place = in_block ? s * stride + (stride - 1 - nw) : s * stride + nd;
					dia_off[place] = (int32_t)off;
					dia_val[place] = v0;
				}
				if (in_block)
					nw++;
				else
					nd++;
				q++;
			}
		}
		while (q < end) {
			LPP_KEEP_ENTRY(q);
			q++;
		}
		if (!FILL) {
			if (valid) rest_len[row] = (end - pbeg) - nd - nw - ndg;
			if (valid && xdiag && ndg == 0) local_nodiag++;
			local_max = max(local_max, (unsigned long long)(nd + nw + 1));
			local_sum += (unsigned long long)(nd + nw);
		}
	}
#undef LPP_KEEP_ENTRY
#undef LPP_COL
#undef LPP_VAL
	if (!FILL) {
		if (lane == 0) {
			atomicMax(&stats[0], local_max);
			atomicAdd(&stats[1], local_sum);
		}
		if (local_nodiag) atomicAdd(&stats[2], local_nodiag);
	}
}

// inverse (lpp_engine_get_csr): merge a row's rest entries with its slice's shared entries (both groups) and its
// diagonal code by column
template <typename T>
__global__ void k_dia_merge(SliceGeom g, const int64_t* __restrict__ rowptr, const int64_t* __restrict__ rrowptr,
                            const int32_t* __restrict__ rcol, const T* __restrict__ rval, int stride,
                            const int32_t* __restrict__ dia_off, const T* __restrict__ dia_val, int32_t* __restrict__ col_out,
                            T* __restrict__ val_out, const uint8_t* __restrict__ dcode, const double* __restrict__ dict)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= g.nrows) return;
	const int64_t blk = r / g.B;
	const int64_t s = blk * g.spb + (r - blk * g.B) / 64;
	int64_t i = rrowptr[r], iend = rrowptr[r + 1], o = rowptr[r];
	int64_t d = s * stride, w = s * stride + stride - 1; // global group ascends from the front, window group from the back
	const int64_t dlim = s * stride + stride, wlim = s * stride;
	bool hg = dcode != nullptr; // the row's diagonal, when it was split off
	while (true) {
		const bool hd = d < dlim && dia_off[d] != kDiaNone, hw = stride > 0 && w >= wlim && w >= d && dia_off[w] != kDiaNone, hi = i < iend;
		if (!hd && !hw && !hi && !hg) break;
		const int64_t cd = hd ? r + (int64_t)dia_off[d] : INT64_MAX;
		const int64_t cw = hw ? r + (int64_t)dia_off[w] : INT64_MAX;
		const int64_t ci = hi ? (int64_t)rcol[i] : INT64_MAX;
		const int64_t cg = hg ? r : INT64_MAX;
		const int64_t c = min(min(cd, cw), min(ci, cg));
		col_out[o] = (int32_t)c;
		if (c == cg) {
			const uint32_t cc = sizeof(T) == 16 ? (uint32_t)((const uint16_t*)dcode)[r] : (uint32_t)dcode[r];
			val_out[o] = CodeTraits<T>::decode(cc, 0, dict);
			hg = false;
		} else if (c == cd) {
			val_out[o] = dia_val[d];
			d++;
		} else if (c == cw) {
			val_out[o] = dia_val[w];
			w--;
		} else {
			val_out[o] = rval[i];
			i++;
		}
		o++;
	}
}

// Block-periodic structure (16-bit block-local columns only): *differs = 1 unless every row block has the row lengths
// and the local column stream of block 0.  One wave per slice of blocks 1..nblocks-1.
// differs[1] = 1 unless the code words repeat too.
static __global__ __launch_bounds__(kBlock) void k_tmpl_check(SliceGeom g, const int64_t* __restrict__ slice_ptr,
                                                               const int32_t* __restrict__ row_len, const uint16_t* __restrict__ col16,
                                                               const int64_t* __restrict__ code_ptr, const uint32_t* __restrict__ codes,
                                                               int* differs)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	bool bad = false, badc = false;
	for (int64_t s = g.spb + wave0; s < g.nslices; s += nwaves) {
		const int64_t blk = s / g.spb, j = s - blk * g.spb;
		const int64_t b0 = slice_ptr[j], n0 = slice_ptr[j + 1] - b0, b1 = slice_ptr[s], n1 = slice_ptr[s + 1] - b1;
		if (n0 != n1) {
			bad = true;
			continue;
		}
		const int64_t r = j * 64 + lane;
		if (r < g.B) bad |= row_len[blk * g.B + r] != row_len[r];
		for (int64_t i = lane; i < n0; i += 64) bad |= col16[b1 + i] != col16[b0 + i];
		const int64_t c0 = code_ptr[j], m0 = code_ptr[j + 1] - c0, c1 = code_ptr[s], m1 = code_ptr[s + 1] - c1;
		if (m0 != m1) {
			badc = true;
			continue;
		}
		for (int64_t i = lane; i < m0; i += 64) badc |= codes[c1 + i] != codes[c0 + i];
	}
	if (bad) differs[0] = 1;
	if (badc) differs[1] = 1;
}

// Packed padded copy of a level-2 template (see SlicedArgs::tw).  One wave per slice of block 0.
// PASS 0: tw_len[j] = padded slots of slice j;  PASS 1 (after the scan of 64*tw_len into tw_off): the words.
template <typename T, int PASS>
__global__ __launch_bounds__(kBlock) void k_tmpl_pack(SliceGeom g, const int64_t* __restrict__ slice_ptr, const int32_t* __restrict__ row_len,
                                                       const uint16_t* __restrict__ col16, const int64_t* __restrict__ code_ptr,
                                                       const uint32_t* __restrict__ codes, int32_t* __restrict__ tw_len,
                                                       const int32_t* __restrict__ tw_off, uint32_t* __restrict__ tw)
{
	constexpr int SPW = CodeTraits<T>::kSlotsPerWord, BITS = CodeTraits<T>::kBits;
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	for (int64_t j = wave0; j < g.spb; j += nwaves) {
		const int64_t r = j * 64 + lane;
		const int len = r < g.B ? row_len[r] : 0;
		int maxlen = len;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
		const int ml = (maxlen + 7) & ~7;
		if (PASS == 0) {
			if (lane == 0) tw_len[j] = ml;
			continue;
		}
		int64_t base = slice_ptr[j];
		const int64_t cbase = code_ptr[j];
		uint32_t* out = tw + tw_off[j];
		const uint32_t own = (uint32_t)min(r, g.B - 1); // padding gathers the row's own window element (times +0.0)
		for (int k = 0; k < ml; k++) {
			const bool on = len > k;
			const unsigned long long m = __ballot(on);
			const int pos = __popcll(m & ((1ull << lane) - 1ull));
			uint32_t w = own;
			if (on) {
				const uint32_t cw = codes[cbase + ((int64_t)(k / SPW) << 6) + lane];
				const uint32_t c = (cw >> (BITS * (k % SPW))) & ((1u << BITS) - 1u);
				w = (uint32_t)col16[base + pos] | (c << 16);
			}
			out[(int64_t)k * 64 + lane] = w;
			base += __popcll(m);
		}
	}
}

// *inside += number of entries whose column lies inside the row's own block of B rows (is an LDS window worth it?)
static __global__ void k_count_local(int64_t nrows, int64_t B, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                     unsigned long long* inside)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long n = 0;
	if (r < nrows) {
		const int64_t r0 = (r / B) * B, r1 = r0 + B;
		for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) n += (col[p] >= r0 && col[p] < r1) ? 1u : 0u;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
	if ((threadIdx.x & 63) == 0 && n) atomicAdd(inside, n);
}

// *outside = 1 when some entry's column lies outside its row block [blk*B, (blk+1)*B)
static __global__ void k_cols_local(SliceGeom g, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int* outside)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= g.nrows) return;
	const int64_t r0 = (r / g.B) * g.B, r1 = r0 + g.B;
	bool bad = false;
	for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) bad |= col[p] < r0 || col[p] >= r1;
	if (bad) *outside = 1;
}

// ---- value dictionary ---------------------------------------------------------------------------
constexpr int kDictTable = 4096; // open-addressing table of distinct 64-bit patterns
constexpr unsigned long long kDictEmpty = ~0ull;

__device__ __forceinline__ unsigned dict_hash(unsigned long long k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	return (unsigned)k & (kDictTable - 1);
}

// ---------------------------------------------------------------------------------------------
// Split-panel layout of a plain-format matrix with a basis block B (no value compression, no model knowledge): every row's
// entries inside its own row block [b*B, (b+1)*B) stay with the LDS-window kernel; the entries that LEAVE the block go into a
// second CSR whose rows are reordered PANEL-major,
//     r' = (panel p, block b, position i)   for row r = b*B + 16p + i,
// i.e. 16 consecutive positions of EVERY block, then the next 16.  In a product basis the leaving entries connect position i of
// block b with position i of other blocks, so while a panel is walked every gather lands in the same 16 positions of all blocks
// (nblocks * 128 bytes = 1.6 MB at BASELINE config 2) and stays in one XCD's L2 -- read in block order those ~17 other blocks'
// rows came from HBM every time (23 of the 102 GB the plain kernel moved).  Bytes stored are those of the CSR plus 4 per row
// (the row map) and a second row-pointer array; nothing is compressed.
// ---------------------------------------------------------------------------------------------

// r' of row r (B = rows per block, nb = whole blocks; rows beyond nb*B -- none for a product basis -- keep their place)
__host__ __device__ inline int64_t panel_major_row(int64_t r, int64_t B, int64_t nb)
{
	if (r >= nb * B) return r;
	const int64_t b = r / B, i = r - b * B, p = i >> 4, npan = (B + 15) >> 4;
	const int64_t wlast = B - (npan - 1) * 16; // positions of the last panel
	const int64_t before = p * 16 * nb; // rows of the panels in front (all of them full)
	const int64_t w = p == npan - 1 ? wlast : 16;
	return before + b * w + (i - p * 16);
}

constexpr int kSplitMaxParts = 4;

// The leaving entries are held in `nparts` CSRs by SOURCE block range (part q: source blocks [q*pblk, (q+1)*pblk)): while a
// panel is walked the gathers of one part touch nblocks/nparts lines (two per source block and panel when rows are not
// 128-byte aligned), which has to stay in L2 next to the entry stream that passes meanwhile.
struct SplitParams {
	int64_t nrows, B, nb;
	int nparts;
	int64_t pblk; // source blocks per part
};
__host__ __device__ inline int split_part_of(const SplitParams& P, int64_t col)
{
	const int64_t q = (col / P.B) / P.pblk;
	return (int)(q < P.nparts ? q : P.nparts - 1);
}

// pass 1: len_in[r], len_out[q][r'] (+ rowmap[r'] = r).  Rows must be sorted by column.
static __global__ void k_split_count(SplitParams P, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t* __restrict__ len_in,
                                     int64_t* const* __restrict__ len_out, int32_t* __restrict__ rowmap)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= P.nrows) return;
	const int64_t lo = (r / P.B) * P.B, hi = lo + P.B;
	int64_t nin = 0, nout[kSplitMaxParts] = { 0, 0, 0, 0 };
	for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
		const int64_t c = col[p];
		if (c >= lo && c < hi) nin++;
		else nout[split_part_of(P, c)]++;
	}
	const int64_t rp = panel_major_row(r, P.B, P.nb);
	len_in[r] = nin;
	for (int q = 0; q < P.nparts; q++) len_out[q][rp] = nout[q];
	rowmap[rp] = (int32_t)r;
}

struct SplitOut {
	const int64_t* rp[kSplitMaxParts];
	int32_t* col[kSplitMaxParts];
	void* val[kSplitMaxParts];
};

// pass 2: the CSRs (row pointers already scanned); columns stay global in all of them
template <typename T>
__global__ void k_split_fill(SplitParams P, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
                             const int64_t* __restrict__ rp_in, int32_t* __restrict__ col_in, T* __restrict__ val_in, SplitOut O)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= P.nrows) return;
	const int64_t lo = (r / P.B) * P.B, hi = lo + P.B;
	const int64_t rpm = panel_major_row(r, P.B, P.nb);
	int64_t qi = rp_in[r], qo[kSplitMaxParts];
	for (int q = 0; q < P.nparts; q++) qo[q] = O.rp[q][rpm];
	for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
		const int64_t c = col[p];
		if (c >= lo && c < hi) {
			col_in[qi] = col[p];
			val_in[qi] = val[p];
			qi++;
		} else {
			const int q = split_part_of(P, c);
			O.col[q][qo[q]] = col[p];
			((T*)O.val[q])[qo[q]] = val[p];
			qo[q]++;
		}
	}
}

// back to one CSR in the given order (lpp_engine_get_csr): every piece of a row is sorted by column and the pieces cover
// disjoint column ranges, except that the in-block run sits inside the range of one part -- a plain merge by column
template <typename T>
__global__ void k_split_merge(SplitParams P, const int64_t* __restrict__ rp_in, const int32_t* __restrict__ col_in, const T* __restrict__ val_in, SplitOut O,
                              const int64_t* __restrict__ rowptr, int32_t* __restrict__ col, T* __restrict__ val)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= P.nrows) return;
	const int64_t rpm = panel_major_row(r, P.B, P.nb);
	int64_t h[kSplitMaxParts + 1], e[kSplitMaxParts + 1];
	h[0] = rp_in[r];
	e[0] = rp_in[r + 1];
	for (int q = 0; q < P.nparts; q++) {
		h[q + 1] = O.rp[q][rpm];
		e[q + 1] = O.rp[q][rpm + 1];
	}
	for (int64_t out = rowptr[r]; out < rowptr[r + 1]; out++) {
		int best = -1;
		int32_t bc = 0;
		for (int q = 0; q <= P.nparts; q++) {
			if (h[q] >= e[q]) continue;
			const int32_t c = q == 0 ? col_in[h[0]] : O.col[q - 1][h[q]];
			if (best < 0 || c < bc) {
				best = q;
				bc = c;
			}
		}
		col[out] = bc;
		val[out] = best == 0 ? val_in[h[0]] : ((const T*)O.val[best - 1])[h[best]];
		h[best]++;
	}
}

// lengths of the merged rows
static __global__ void k_split_lengths(SplitParams P, const int64_t* __restrict__ rp_in, SplitOut O, int64_t* __restrict__ len)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= P.nrows) return;
	const int64_t rpm = panel_major_row(r, P.B, P.nb);
	int64_t n = rp_in[r + 1] - rp_in[r];
	for (int q = 0; q < P.nparts; q++) n += O.rp[q][rpm + 1] - O.rp[q][rpm];
	len[r] = n;
}

// collect the distinct doubles of vals[0..n) into table (pre-filled with kDictEmpty); *overflow != 0 when full
static __global__ __launch_bounds__(kBlock) void k_dict_collect(const double* __restrict__ vals, int64_t n,
                                                                 unsigned long long* table, int* overflow)
{
	unsigned long long last = kDictEmpty;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
		const unsigned long long key = (unsigned long long)__double_as_longlong(vals[i]);
		if (key == last) continue; // runs of equal values are the common case
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

// code of v in the sorted dictionary (bit patterns compared as unsigned integers)
__device__ __forceinline__ uint32_t dict_code(const double* dict, int ndict, double v)
{
	const unsigned long long key = (unsigned long long)__double_as_longlong(v);
	int lo = 0, hi = ndict - 1;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if ((unsigned long long)__double_as_longlong(dict[mid]) < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	return (uint32_t)lo;
}

// one wave per slice: pack the codes of the slice's values (read in CSR order) into the padded word layout
template <typename T>
__global__ __launch_bounds__(kBlock) void k_slice_codes(SliceGeom g, const int64_t* __restrict__ rowptr,
                                                         const T* __restrict__ val_in, const int64_t* __restrict__ code_ptr,
                                                         const double* __restrict__ dict, int ndict,
                                                         uint32_t* __restrict__ codes)
{
	constexpr int SPW = CodeTraits<T>::kSlotsPerWord;
	constexpr int BITS = CodeTraits<T>::kBits;
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	for (int64_t s = wave0; s < g.nslices; s += nwaves) {
		int64_t row0;
		int nvalid;
		slice_rows(g, s, row0, nvalid);
		int64_t p0 = 0;
		int len = 0;
		if (lane < nvalid) {
			p0 = rowptr[row0 + lane];
			len = (int)(rowptr[row0 + lane + 1] - p0);
		}
		int maxlen = len;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
		const int nwords = ((maxlen + 7) / 8) * (8 / SPW);
		const int64_t cbase = code_ptr[s];
		for (int w = 0; w < nwords; w++) {
			uint32_t word = 0;
#pragma unroll
			for (int q = 0; q < SPW; q++) {
				const int k = w * SPW + q;
				if (k < len) {
					const double* pv = (const double*)(val_in + p0 + k);
					uint32_t c = dict_code(dict, ndict, pv[0]);
					if (sizeof(T) == 16) c |= dict_code(dict, ndict, pv[1]) << 8;
					word |= c << (BITS * q);
				}
			}
			codes[cbase + ((int64_t)w << 6) + lane] = word;
		}
	}
}

// decode back to CSR order (for lpp_engine_get_csr)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_slice_decode(SliceGeom g, const int64_t* __restrict__ rowptr,
                                                          const uint32_t* __restrict__ codes, const int64_t* __restrict__ code_ptr,
                                                          const double* __restrict__ dict, T* __restrict__ val_out, int tmpl_codes = 0)
{
	constexpr int SPW = CodeTraits<T>::kSlotsPerWord;
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	for (int64_t s = wave0; s < g.nslices; s += nwaves) {
		int64_t row0;
		int nvalid;
		slice_rows(g, s, row0, nvalid);
		if (lane >= nvalid) continue;
		const int64_t p0 = rowptr[row0 + lane];
		const int len = (int)(rowptr[row0 + lane + 1] - p0);
		const int64_t cbase = code_ptr[tmpl_codes ? s % g.spb : s];
		for (int k = 0; k < len; k++) val_out[p0 + k] = CodeTraits<T>::decode(codes[cbase + ((int64_t)(k / SPW) << 6) + lane], k % SPW, dict);
	}
}

} // namespace lpp
