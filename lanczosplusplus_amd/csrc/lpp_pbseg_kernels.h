// lpp_pbseg_kernels.h -- k_pb_up_seg: the in-block part  u = alpha (T y + D y)  of a product-basis matrix for rows beyond one LDS
// window, with T decomposed by the high sites of the species' basis word (lpp_pbseg.h).  Replaces k_pb_up_big2 where the host finds
// the structure (pb_seg_plan); same vectors, same coupling kernel, same streaming pass.
//
// Work item = (pair of blocks of one XCD, item): the item's run of stored positions of both blocks is staged in two LDS windows,
// every word loaded is applied to both blocks.  Per slice of 64 positions (one segment):
//   low-low hops   LDS gathers through the item TYPE's 16-bit lists (k_pb_up's format; ~0.8 MB for all types together: L2-resident),
//   cross hops     one 16-bit word per position and hop from the class's table (L2-resident) + one 8-byte read from the row in memory,
//   high-high hops one coalesced 512-byte read per hop, no words,
// the per-segment scalars (source segment, value, table) sit in LDS.  What is read from the row in memory belongs to the SAME two
// blocks the XCD's other workgroups are staging at that moment: L2 hits.
#pragma once
#include "lpp_pb_kernels.h"
#include "lpp_pbseg.h"

namespace lpp {

struct PbSegArgs {
	const SegItem* items;
	int nitems;
	const SegCross* cross; // [segment][NC]
	const SegHh* hh; // [segment][NH]
	const SegSlice* slices;
	const uint32_t* tw;
	const uint32_t* xw; // two 16-bit cross words per position
	int G;
	double gval[2];
	const double* dict; // 256 doubles (diagonal codes)
	const uint8_t* dcode; // one code per row, pitched like the vectors
	int64_t n_up, pitch, n_blk;
	int ws; // distance of the two windows in elements
	int wmax; // longest item
	const double* y;
	double* u;
	double* partial; // per-workgroup Re<y|u> (null: not wanted)
	EpiScale sc; // only alpha is used
	int flat; // 1: work items are dealt over ALL workgroups (few blocks: a Heisenberg chain is one block), 0: blocks by XCD
};

constexpr int kSegThreads = 512; // 8 waves of up to 256 registers: two slices' loads from the rows in flight per wave
constexpr int kSegPre = 4; // in-window chunks of every (slice, group) requested one slice ahead
constexpr int kSegStage = 16; // 16-byte pieces per thread that stage the two windows: ALL in flight together (2 x 8128 elements at most)

// LDS: [0, 2 ws) windows | dcode[2][wmax + 32] | slice heads[160] | cross[16 * 12] | hh[16 * 8] | dict[256] | smem[8]
__host__ __device__ inline size_t pb_seg_dcode_stride(int wmax) { return ((size_t)wmax + 32 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pb_seg_tab_offset(int ws, int wmax) { return (2 * sizeof(double) * (size_t)ws + 2 * pb_seg_dcode_stride(wmax) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pb_seg_lds_bytes(int ws, int wmax)
{
	return pb_seg_tab_offset(ws, wmax) + sizeof(SegSlice) * kSegMaxSlices + sizeof(SegCross) * kSegMaxSegs * kSegMaxCross + sizeof(SegHh) * kSegMaxSegs * kSegMaxHh + sizeof(double) * (256 + kSegThreads / 64) + 16;
}

template <int GG> struct SegHeads { // wave-uniform
	int nc[GG], off[GG];
	int first, count, segoff; // the slice
	int seg; // its segment (of the item)
};
template <int GG> struct SegWinWords {
	uint4 w[GG][kSegPre / 2]; // two chunks per 16-byte load
};
template <int NC> struct SegCrossWords {
	uint32_t x[NC]; // two 16-bit words each
};
template <int NC, int NH, int ROWS> struct SegData { // what a slice reads from the rows in memory, both blocks
	double xa[2 * NC], xb[ROWS == 2 ? 2 * NC : 1], ha[NH], hb[ROWS == 2 ? NH : 1];
	uint32_t sg; // bits 14..15 of the cross words, two bits per hop
};

// GT value groups (1 or 2); P0 / P1: chunks of group 0 / 1 requested ahead (lists beyond that are streamed: rare by the host's choice);
// NC / NH: PAIRS of cross hops / high-high hops per segment -- every segment's lists are padded to that with entries of value 0.0 (pb_seg_plan), so the
// slice loop has no condition on them and the compiler's s_waitcnt counts are exact: every vector-memory load of the slice loop is issued
// a whole slice before it is used, the data of slice m+1 before the gathers of slice m (results return in order).  Nothing in the slice
// loop is a scalar load either (their out-of-order return makes every wait for an LDS gather a wait for all of them): the slice heads
// and the segments' scalars are LDS copies made while the windows are staged, and that staging is ONE round trip -- all of a thread's
// pieces of both windows, the codes and the tables are requested before the first is stored (the first version's 6 + 6 round trips per
// item were a third of its time).
// ROWS: blocks per workgroup (2: every word loaded serves two blocks; 1: a matrix of one block -- a Heisenberg chain, whose S+S- part
// is the hopping matrix of its up spins (pb_chain) -- where the second window stays unused).
template <bool DOT, int GT, int P0, int P1, int NC, int NH, int ROWS = 2> __global__ __launch_bounds__(kSegThreads) void k_pb_up_seg(PbSegArgs a)
{
	static_assert(GT == 1 || GT == 2, "one or two value groups");
	static_assert(ROWS == 1 || ROWS == 2, "one or two blocks per workgroup");
	static_assert(P0 <= kSegPre && P1 <= kSegPre && !(P0 & 1) && !(P1 & 1) && NC <= kSegMaxCross && NH <= kSegMaxHh && NC <= 8, "limits");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // block 0's window at LDS address 0 (a list entry * 8 IS the byte address), block 1's at ws * 8
	const int WS = a.ws;
	const size_t dstride = pb_seg_dcode_stride(a.wmax);
	uint8_t* dcode_s = (uint8_t*)(win + 2 * WS); // [2][dstride]
	SegSlice* heads_s = (SegSlice*)(lds_raw + pb_seg_tab_offset(a.ws, a.wmax));
	SegCross* cross_s = (SegCross*)(heads_s + kSegMaxSlices);
	SegHh* hh_s = (SegHh*)(cross_s + kSegMaxSegs * kSegMaxCross);
	double* dict_s = (double*)(hh_s + kSegMaxSegs * kSegMaxHh);
	double* smem = dict_s + 256;
	for (int i = threadIdx.x; i < 256; i += kSegThreads) dict_s[i] = a.dict[i];
	double alpha, beta_unused;
	epi_coeffs(a.sc, alpha, beta_unused);
	constexpr int NW = kSegThreads / 64;
	constexpr int GG = GT;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint4* const tw4 = (const uint4*)a.tw;
	const uint32_t* const xw = a.xw;
	const int nitems = a.nitems;
	const uint32_t wbytes = (uint32_t)WS * 8u;
	double* const uout = a.u;
	double gv[GG];
#pragma unroll
	for (int g = 0; g < GG; g++) gv[g] = a.gval[g];
	double dot = 0.0;
	// blocks xcd, xcd + 8, ... belong to this workgroup's XCD (round-robin dispatch: speed only); its workgroups walk
	// seq = (pair of blocks, item) together, so the rows that are read from memory are the ones being staged right now
	const int nx = (gridDim.x & 7) == 0 && !a.flat ? 8 : 1;
	const int xcd = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int64_t slot = nx == 8 ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x, nslots = gridDim.x / nx;
	const int64_t nbx = (a.n_blk - xcd + nx - 1) / nx;
	const int64_t npairs = ROWS == 2 ? (nbx + 1) >> 1 : nbx;
	auto gather4x2 = [=](const uint2& w, double& a0, double& a1, double& b0, double& b1) __attribute__((always_inline)) {
		const uint32_t p0 = pb_lo8(w.x), p1 = pb_hi8(w.x), p2 = pb_lo8(w.y), p3 = pb_hi8(w.y);
		a0 += pb_lds_abs(p0);
		if (ROWS == 2) b0 += pb_lds_abs(p0 + wbytes);
		a1 += pb_lds_abs(p1);
		if (ROWS == 2) b1 += pb_lds_abs(p1 + wbytes);
		a0 += pb_lds_abs(p2);
		if (ROWS == 2) b0 += pb_lds_abs(p2 + wbytes);
		a1 += pb_lds_abs(p3);
		if (ROWS == 2) b1 += pb_lds_abs(p3 + wbytes);
	};
	for (int64_t seq = slot; seq < npairs * nitems; seq += nslots) {
		const int64_t pr = seq / nitems;
		const int it = (int)(seq - pr * nitems);
		const int64_t blk0 = (ROWS * pr) * nx + xcd;
		const bool two = ROWS == 2 && 2 * pr + 1 < nbx; // the last pair of an odd count holds one block: its twin re-reads it and stores nothing
		const int64_t blk1 = two ? (2 * pr + 1) * nx + xcd : blk0;
		const SegItem I = a.items[it];
		const int c0 = I.c0, wlen = I.wlen, nsl = I.nslices;
		const int64_t rowbase0 = blk0 * a.pitch, rowbase1 = blk1 * a.pitch;
		const double* const yrow0 = a.y + rowbase0;
		const double* const yrow1 = a.y + rowbase1;
		{
			// the run [c0, c0 + wlen) starts at any element: the pairs [e0, e1) around it are loaded as aligned 16-byte pieces and land at
			// window index e - c0 + kSegWinPad (>= 1), as two 8-byte LDS stores (the pair may straddle a 16-byte LDS boundary).
			// Everything this thread stages is requested here, in front of the barrier that waits for the previous item's slices
			const int e0 = c0 & ~1, e1 = (c0 + wlen + 1) & ~1;
			const int p2 = (e1 - e0) >> 1, lbase = e0 - c0 + kSegWinPad;
			constexpr int NST = kSegStage * ROWS / 2;
			double2 t[NST];
#pragma unroll
			for (int k = 0; k < NST; k++) { // [0, p2): block 0, [p2, 2 p2): block 1; clamped lanes re-load the last pair
				const int idx = min((int)threadIdx.x + k * kSegThreads, ROWS * p2 - 1);
				const bool second = idx >= p2;
				t[k] = ((const double2*)((second ? yrow1 : yrow0) + e0))[second ? idx - p2 : idx];
			}
			const int d0 = c0 & ~15, p16 = (((c0 + wlen + 15) & ~15) - d0) >> 4; // 16 codes per piece: at most 2 x 510 pieces
			uint4 dc[ROWS];
#pragma unroll
			for (int k = 0; k < ROWS; k++) {
				const int i0 = min((int)threadIdx.x + k * kSegThreads, ROWS * p16 - 1);
				const bool second = i0 >= p16;
				dc[k] = ((const uint4*)(a.dcode + (second ? rowbase1 : rowbase0) + d0))[second ? i0 - p16 : i0];
			}
			const int nhd = nsl, ncr = I.nseg * NC * 2, nhh = I.nseg * NH; // <= 160, 192, 128: one 16-byte piece per thread each
			const uint4 hd = ((const uint4*)(a.slices + I.slice_first))[min((int)threadIdx.x, nhd - 1)];
			const uint4 cr = ((const uint4*)(a.cross + (size_t)I.seg_first * NC))[min((int)threadIdx.x, ncr - 1)]; // (an entry is two pieces)
			const uint4 hq = ((const uint4*)(a.hh + (size_t)I.seg_first * NH))[min((int)threadIdx.x, nhh - 1)];
			__syncthreads(); // everyone is done with the previous windows and tables
#pragma unroll
			for (int k = 0; k < NST; k++) {
				const int idx = min((int)threadIdx.x + k * kSegThreads, ROWS * p2 - 1);
				const bool second = idx >= p2;
				double* const d = win + (second ? WS : 0) + lbase + 2 * (second ? idx - p2 : idx);
				d[0] = t[k].x;
				d[1] = t[k].y;
			}
#pragma unroll
			for (int k = 0; k < ROWS; k++) {
				const int i0 = min((int)threadIdx.x + k * kSegThreads, ROWS * p16 - 1);
				const bool second = i0 >= p16;
				((uint4*)(dcode_s + (second ? dstride : 0)))[second ? i0 - p16 : i0] = dc[k];
			}
			if ((int)threadIdx.x < nhd) ((uint4*)heads_s)[threadIdx.x] = hd;
			if ((int)threadIdx.x < ncr) ((uint4*)cross_s)[threadIdx.x] = cr;
			if ((int)threadIdx.x < nhh) ((uint4*)hh_s)[threadIdx.x] = hq;
			if (threadIdx.x < ROWS * kPbZeroSlots) win[(threadIdx.x >> 5) * WS + I.zero_at + (threadIdx.x & 31)] = 0.0;
		}
		__syncthreads();
		const int doff = (c0 & 15); // dcode_s index of the item's first position
		auto load_heads = [=](int jj, SegHeads<GG>& h) __attribute__((always_inline)) {
			const uint4 q = ((const uint4*)heads_s)[min(jj, nsl - 1)]; // one LDS read, the same address in every lane; beyond the item: a valid slice, never used
			h.first = __builtin_amdgcn_readfirstlane((int)(q.x & 0xffffu));
			h.count = __builtin_amdgcn_readfirstlane((int)((q.x >> 16) & 0xffu));
			h.seg = __builtin_amdgcn_readfirstlane((int)(q.x >> 24));
			h.segoff = __builtin_amdgcn_readfirstlane((int)(q.y & 0xffffu));
			h.nc[0] = __builtin_amdgcn_readfirstlane((int)((q.y >> 16) & 0xffu));
			h.off[0] = __builtin_amdgcn_readfirstlane((int)q.z);
			if (GG == 2) {
				h.nc[GG - 1] = __builtin_amdgcn_readfirstlane((int)(q.y >> 24));
				h.off[GG - 1] = __builtin_amdgcn_readfirstlane((int)q.w);
			}
		};
		auto load_win_words = [=](const SegHeads<GG>& h, SegWinWords<GG>& s) __attribute__((always_inline)) {
#pragma unroll
			for (int g = 0; g < GG; g++) {
				const uint4* wp = tw4 + (size_t)h.off[g] * 64 + lane;
#pragma unroll
				for (int c = 0; c < kSegPre / 2; c++)
					if (2 * c < (g == 0 ? P0 : P1)) s.w[g][c] = wp[c * 64]; // chunks beyond the list belong to the next list (or the slack): never used
			}
		};
		auto load_cross_words = [=](const SegHeads<GG>& h, SegCrossWords<NC>& s) __attribute__((always_inline)) {
#pragma unroll
			for (int b = 0; b < NC; b++) s.x[b] = xw[(size_t)(cross_s[h.seg * NC + b].wordoff + h.segoff) + lane];
		};
		auto issue_data = [=](const SegHeads<GG>& h, const SegCrossWords<NC>& s, SegData<NC, NH, ROWS>& d) __attribute__((always_inline)) {
			uint32_t sg = 0;
#pragma unroll
			for (int b = 0; b < NC; b++) {
				const uint32_t sb = (uint32_t)cross_s[h.seg * NC + b].srcbase;
				const uint32_t at0 = (sb + (s.x[b] & 0x1fffu)) * 8u, at1 = (sb + ((s.x[b] >> 16) & 0x1fffu)) * 8u;
				d.xa[2 * b] = *(const double*)((const char*)yrow0 + at0);
				if (ROWS == 2) d.xb[2 * b] = *(const double*)((const char*)yrow1 + at0);
				d.xa[2 * b + 1] = *(const double*)((const char*)yrow0 + at1);
				if (ROWS == 2) d.xb[2 * b + 1] = *(const double*)((const char*)yrow1 + at1);
				sg |= (((s.x[b] >> 14) & 3u) | ((s.x[b] >> 28) & 0xcu)) << (4 * b);
			}
			d.sg = sg;
			const int lc = min(lane, h.count - 1);
#pragma unroll
			for (int b = 0; b < NH; b++) {
				const uint32_t at = (uint32_t)(hh_s[h.seg * NH + b].srcbase + (h.segoff + lc) * hh_s[h.seg * NH + b].pad) * 8u;
				d.ha[b] = *(const double*)((const char*)yrow0 + at);
				if (ROWS == 2) d.hb[b] = *(const double*)((const char*)yrow1 + at);
			}
		};
		auto compute = [=, &dot](int jj, const SegHeads<GG>& h, const SegWinWords<GG>& s, const SegData<NC, NH, ROWS>& d) __attribute__((always_inline)) {
			if (jj >= nsl) return; // wave-uniform
			double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
			for (int g = 0; g < GG; g++) {
				const int nc = h.nc[g];
				constexpr int kDepth[2] = { P0, P1 };
				const int depth = kDepth[g];
				double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
				for (int c = 0; c < kSegPre; c += 2) { // pairs of chunks (depth is even)
					if (c < depth) {
						const uint4 q = s.w[g][c >> 1];
						if (nc >= c + 2) {
							gather4x2(uint2 { q.x, q.y }, a0, a1, b0, b1);
							gather4x2(uint2 { q.z, q.w }, a0, a1, b0, b1);
						} else if (nc == c + 1) {
							gather4x2(uint2 { q.x, q.y }, a0, a1, b0, b1);
						}
					}
				}
				if (nc > depth) { // longer lists: the rest streamed
					const uint4* wp = tw4 + (size_t)h.off[g] * 64 + lane;
					for (int c = depth; c < nc; c += 2) {
						const uint4 q = wp[(c >> 1) * 64];
						gather4x2(uint2 { q.x, q.y }, a0, a1, b0, b1);
						if (c + 1 < nc) gather4x2(uint2 { q.z, q.w }, a0, a1, b0, b1);
					}
				}
				acc0 = fma(gv[g], a0 + a1, acc0);
				if (ROWS == 2) acc1 = fma(gv[g], b0 + b1, acc1);
			}
#pragma unroll
			for (int b = 0; b < 2 * NC; b++) {
				// two bits per hop as a signed field: +1, -1 or 0 (no entry: the element read was the source segment's first)
				const double v = cross_s[h.seg * NC + (b >> 1)].val[b & 1] * (double)((int32_t)(d.sg << (30 - 2 * b)) >> 30);
				acc0 = fma(v, d.xa[b], acc0);
				if (ROWS == 2) acc1 = fma(v, d.xb[b], acc1);
			}
#pragma unroll
			for (int b = 0; b < NH; b++) {
				const double v = hh_s[h.seg * NH + b].val;
				acc0 = fma(v, d.ha[b], acc0);
				if (ROWS == 2) acc1 = fma(v, d.hb[b], acc1);
			}
			const int lc = min(lane, h.count - 1);
			const int il = h.first + lc; // position in the item
			const double y0 = win[il + kSegWinPad], y1 = ROWS == 2 ? win[WS + il + kSegWinPad] : 0.0;
			acc0 = fma(dict_s[dcode_s[il + doff]], y0, acc0);
			if (ROWS == 2) acc1 = fma(dict_s[dcode_s[dstride + il + doff]], y1, acc1);
			if (lane < h.count) {
				const double u0 = alpha * acc0, u1 = alpha * acc1;
				__builtin_nontemporal_store(u0, &uout[rowbase0 + c0 + il]);
				if (DOT) dot += y0 * u0;
				if (two) {
					__builtin_nontemporal_store(u1, &uout[rowbase1 + c0 + il]);
					if (DOT) dot += y1 * u1;
				}
			}
		};
		// slices m = 0, 1, ... of this wave are jj = wave + m NW.  At the top of an iteration: WA = lists of m, DA = data of m (in flight),
		// XB = cross words of m + 1 (loaded), XA = cross words of m + 2 (in flight); fixed register sets, the loop is unrolled by two
		SegHeads<GG> h0, h1, h2, h3, h4;
		SegWinWords<GG> WA, WB;
		SegCrossWords<NC> XA, XB;
		SegData<NC, NH, ROWS> DA, DB;
		load_heads(wave, h0);
		load_heads(wave + NW, h1);
		load_heads(wave + 2 * NW, h2);
		load_cross_words(h0, XA);
		load_win_words(h0, WA);
		load_cross_words(h1, XB);
		issue_data(h0, XA, DA);
		load_cross_words(h2, XA);
		for (int jj = wave; jj < nsl; jj += 2 * NW) {
			issue_data(h1, XB, DB);
			load_win_words(h1, WB);
			load_heads(jj + 3 * NW, h3);
			load_cross_words(h3, XB);
			compute(jj, h0, WA, DA);
			issue_data(h2, XA, DA);
			load_win_words(h2, WA);
			load_heads(jj + 4 * NW, h4);
			load_cross_words(h4, XA);
			compute(jj + NW, h1, WB, DB);
			h0 = h2;
			h1 = h3;
			h2 = h4;
		}
	}
	if (DOT) {
		const double r = block_sum_n<kSegThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

} // namespace lpp
