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
	// 1: one block only (pb_chain) -- its items are dealt over ALL workgroups, item i to workgroup i mod gridDim.  (Contiguous ranges of the
	// stored order per XCD, so that the segments a slice reads from memory were staged by the same XCD a moment ago, cut the fabric reads
	// from 2.4 to 1.9 GB per product at L = 28 and ran 4 % SLOWER whatever the balance between the ranges: not kept.)  0: blocks by XCD
	int flat;
};

constexpr int kSegThreads = 512; // 8 waves of up to 256 registers: two slices' loads from the rows in flight per wave
constexpr int kSegPre = 4; // in-window chunks of every (slice, group) requested one slice ahead
constexpr int kSegStage = 16; // 16-byte pieces per thread that stage the two windows: ALL in flight together (2 x 8128 elements at most)

// LDS: [0, 2 ws) windows | dcode[2][wmax + 32] | 2 x slice heads[160] | 2 x cross[16 * 6] | 2 x hh[16 * 12] | dict[256] | smem[8] | per wave 64 x 2 doubles (high-high sums)
__host__ __device__ inline size_t pb_seg_dcode_stride(int wmax) { return ((size_t)wmax + 32 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pb_seg_tab_offset(int ws, int wmax) { return (2 * sizeof(double) * (size_t)ws + 2 * pb_seg_dcode_stride(wmax) + 15) & ~(size_t)15; }
// (rows: blocks per workgroup; one block per workgroup alternates between TWO sets of tables)
__host__ __device__ inline size_t pb_seg_lds_bytes(int ws, int wmax, int rows)
{
	return pb_seg_tab_offset(ws, wmax) + (rows == 1 ? 2 : 1) * (sizeof(SegSlice) * kSegMaxSlices + sizeof(SegCross) * kSegMaxSegs * kSegMaxCross + sizeof(SegHh) * kSegMaxSegs * pb_seg_hh_cap(rows)) + sizeof(double) * (256 + kSegThreads / 64 + 2 * kSegThreads) + 16;
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
template <int N> struct SegInt {
	static constexpr int value = N;
};
struct __attribute__((aligned(8))) SegPair8 { // two consecutive elements of a row at any (8-byte aligned) position: one 16-byte load
	double x, y;
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
	static_assert(P0 <= kSegPre && P1 <= kSegPre && !(P0 & 1) && !(P1 & 1) && NC >= 1 && NC <= kSegMaxCross && NH <= pb_seg_hh_cap(ROWS) && NC <= 8 && !(NH & 1), "limits");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // block 0's window at LDS address 0 (a list entry * 8 IS the byte address), block 1's at ws * 8
	const int WS = a.ws;
	const size_t dstride = pb_seg_dcode_stride(a.wmax);
	uint8_t* dcode_s = (uint8_t*)(win + 2 * WS); // [2][dstride]
	SegSlice* heads_s = (SegSlice*)(lds_raw + pb_seg_tab_offset(a.ws, a.wmax));
	constexpr int NT = ROWS == 1 ? 2 : 1; // sets of tables: one block per workgroup alternates between two
	SegCross* cross_s = (SegCross*)(heads_s + NT * kSegMaxSlices);
	SegHh* hh_s = (SegHh*)(cross_s + NT * kSegMaxSegs * kSegMaxCross);
	double* dict_s = (double*)(hh_s + NT * kSegMaxSegs * pb_seg_hh_cap(ROWS));
	double* smem = dict_s + 256;
	double* const tr_s = smem + kSegThreads / 64 + (threadIdx.x >> 6) * 128; // this wave's 64 pairs (compute: the high-high sums change lanes)
	for (int i = threadIdx.x; i < 256; i += kSegThreads) dict_s[i] = a.dict[i];
	double alpha, beta_unused;
	epi_coeffs(a.sc, alpha, beta_unused);
	constexpr int NW = kSegThreads / 64;
	constexpr int GG = GT;
	// one block per workgroup = a chain (pb_chain): a high site has ONE low neighbour, so a "pair" of cross hops holds one hop and the second
	// read of every pair is left out (the planner refuses a one-block plan with a true pair)
	constexpr bool X1 = ROWS == 1;
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
	// Staging of an item: the run [c0, c0 + wlen) starts at any element: the pairs [e0, e1) around it are loaded as aligned 16-byte pieces
	// and land at window index e - c0 + kSegWinPad (>= 1), as two 8-byte LDS stores (the pair may straddle a 16-byte LDS boundary).  ALL
	// of a thread's pieces -- rows, codes, slice heads, the segments' scalars -- are requested before the first is stored: one round trip.
	constexpr int NST = kSegStage * ROWS / 2;
	struct Stage {
		double2 t[NST];
		uint4 dc[ROWS];
		uint4 hd, cr, hq;
	};
	struct Work { // one (blocks, item) of the sequence
		SegItem I;
		int64_t rowbase0, rowbase1;
		bool two;
	};
	auto decode = [=](int64_t sq, Work& w) __attribute__((always_inline)) {
		const int64_t pr = sq / nitems;
		const int it = (int)(sq - pr * nitems);
		const int64_t blk0 = (ROWS * pr) * nx + xcd;
		w.two = ROWS == 2 && 2 * pr + 1 < nbx; // the last pair of an odd count holds one block: its twin re-reads it and stores nothing
		const int64_t blk1 = w.two ? (2 * pr + 1) * nx + xcd : blk0;
		w.I = a.items[it];
		w.rowbase0 = blk0 * a.pitch;
		w.rowbase1 = blk1 * a.pitch;
	};
	auto stage_load = [=](const Work& w, Stage& S) __attribute__((always_inline)) {
		const int c0 = w.I.c0, wlen = w.I.wlen;
		const double* const yr0 = a.y + w.rowbase0;
		const double* const yr1 = a.y + w.rowbase1;
		const int e0 = c0 & ~1, e1 = (c0 + wlen + 1) & ~1;
		const int p2 = (e1 - e0) >> 1;
#pragma unroll
		for (int k = 0; k < NST; k++) { // [0, p2): block 0, [p2, 2 p2): block 1; clamped lanes re-load the last pair
			const int idx = min((int)threadIdx.x + k * kSegThreads, ROWS * p2 - 1);
			const bool second = idx >= p2;
			S.t[k] = ((const double2*)((second ? yr1 : yr0) + e0))[second ? idx - p2 : idx];
		}
		const int d0 = c0 & ~15, p16 = (((c0 + wlen + 15) & ~15) - d0) >> 4; // 16 codes per piece: at most 510 pieces per row
#pragma unroll
		for (int k = 0; k < ROWS; k++) {
			const int i0 = min((int)threadIdx.x + k * kSegThreads, ROWS * p16 - 1);
			const bool second = i0 >= p16;
			S.dc[k] = ((const uint4*)(a.dcode + (second ? w.rowbase1 : w.rowbase0) + d0))[second ? i0 - p16 : i0];
		}
		// <= 160 slice heads, 16 x 6 x 2 cross pieces, 16 x 12 high-high entries: one 16-byte piece per thread each
		S.hd = ((const uint4*)(a.slices + w.I.slice_first))[min((int)threadIdx.x, w.I.nslices - 1)];
		S.cr = ((const uint4*)(a.cross + (size_t)w.I.seg_first * NC))[min((int)threadIdx.x, w.I.nseg * NC * 2 - 1)]; // (an entry is two pieces)
		S.hq = ((const uint4*)(a.hh + (size_t)w.I.seg_first * NH))[min((int)threadIdx.x, w.I.nseg * NH - 1)];
	};
	// ROWS == 2: the two blocks' windows are windows 0 and 1; ROWS == 1: the block's window is window `buf` (with its own codes and tables)
	auto stage_store = [=](const Work& w, int buf, const Stage& S) __attribute__((always_inline)) {
		const int c0 = w.I.c0, wlen = w.I.wlen;
		const int e0 = c0 & ~1, e1 = (c0 + wlen + 1) & ~1;
		const int p2 = (e1 - e0) >> 1, lbase = e0 - c0 + kSegWinPad;
#pragma unroll
		for (int k = 0; k < NST; k++) {
			const int idx = min((int)threadIdx.x + k * kSegThreads, ROWS * p2 - 1);
			const bool second = idx >= p2;
			double* const d = win + (ROWS == 2 ? (second ? WS : 0) : buf * WS) + lbase + 2 * (second ? idx - p2 : idx);
			d[0] = S.t[k].x;
			d[1] = S.t[k].y;
		}
		const int d0 = c0 & ~15, p16 = (((c0 + wlen + 15) & ~15) - d0) >> 4;
#pragma unroll
		for (int k = 0; k < ROWS; k++) {
			const int i0 = min((int)threadIdx.x + k * kSegThreads, ROWS * p16 - 1);
			const bool second = i0 >= p16;
			((uint4*)(dcode_s + (ROWS == 2 ? (second ? dstride : 0) : buf * dstride)))[second ? i0 - p16 : i0] = S.dc[k];
		}
		if ((int)threadIdx.x < w.I.nslices) ((uint4*)(heads_s + buf * kSegMaxSlices))[threadIdx.x] = S.hd;
		if ((int)threadIdx.x < w.I.nseg * NC * 2) ((uint4*)(cross_s + buf * kSegMaxSegs * kSegMaxCross))[threadIdx.x] = S.cr;
		if ((int)threadIdx.x < w.I.nseg * NH) ((uint4*)(hh_s + buf * kSegMaxSegs * pb_seg_hh_cap(ROWS)))[threadIdx.x] = S.hq;
		if (threadIdx.x < ROWS * kPbZeroSlots) win[(ROWS == 2 ? (threadIdx.x >> 5) : buf) * WS + w.I.zero_at + (threadIdx.x & 31)] = 0.0;
	};
	const int64_t total = npairs * nitems;
	Stage S;
	Work wk, wn;
	int cur = 0;
	if (ROWS == 1) { // one block per workgroup: the second window takes the NEXT item while this one's slices run (one barrier per item)
		if (slot < total) {
			decode(slot, wk);
			stage_load(wk, S);
			stage_store(wk, 0, S);
		}
		__syncthreads();
	}
	for (int64_t seq = slot; seq < total; seq += nslots) {
		bool has_next = false;
		if (ROWS == 2) {
			decode(seq, wk);
			stage_load(wk, S); // requested in front of the barrier that waits for the previous item's slices
			__syncthreads(); // everyone is done with the previous windows and tables
			stage_store(wk, 0, S);
			__syncthreads();
		} else {
			has_next = seq + nslots < total;
			if (has_next) {
				decode(seq + nslots, wn);
				stage_load(wn, S); // in flight while this item's slices run
			}
		}
		const int c0 = wk.I.c0, nsl = wk.I.nslices;
		const bool two = wk.two;
		const int64_t rowbase0 = wk.rowbase0, rowbase1 = wk.rowbase1;
		const double* const yrow0 = a.y + rowbase0;
		const double* const yrow1 = a.y + rowbase1;
		// this item's window, codes and tables
		const uint32_t woff = ROWS == 1 ? (uint32_t)cur * wbytes : 0u;
		const double* const win_c = win + (ROWS == 1 ? cur * WS : 0);
		const uint8_t* const dcode_c = dcode_s + (ROWS == 1 ? cur * dstride : 0);
		const SegSlice* const heads_c = heads_s + (ROWS == 1 ? cur * kSegMaxSlices : 0);
		const SegCross* const cross_c = cross_s + (ROWS == 1 ? cur * kSegMaxSegs * kSegMaxCross : 0);
		const SegHh* const hh_c = hh_s + (ROWS == 1 ? cur * kSegMaxSegs * pb_seg_hh_cap(ROWS) : 0);
		auto gather4x2 = [=](const uint2& w, double& a0, double& a1, double& b0, double& b1) __attribute__((always_inline)) {
			const uint32_t p0 = pb_lo8(w.x) + woff, p1 = pb_hi8(w.x) + woff, p2 = pb_lo8(w.y) + woff, p3 = pb_hi8(w.y) + woff;
			a0 += pb_lds_abs(p0);
			if (ROWS == 2) b0 += pb_lds_abs(p0 + wbytes);
			a1 += pb_lds_abs(p1);
			if (ROWS == 2) b1 += pb_lds_abs(p1 + wbytes);
			a0 += pb_lds_abs(p2);
			if (ROWS == 2) b0 += pb_lds_abs(p2 + wbytes);
			a1 += pb_lds_abs(p3);
			if (ROWS == 2) b1 += pb_lds_abs(p3 + wbytes);
		};
		const int doff = (c0 & 15); // dcode_s index of the item's first position
		auto load_heads = [=](int jj, SegHeads<GG>& h) __attribute__((always_inline)) {
			const uint4 q = ((const uint4*)heads_c)[min(jj, nsl - 1)]; // one LDS read, the same address in every lane; beyond the item: a valid slice, never used
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
			for (int b = 0; b < NC; b++) s.x[b] = xw[(size_t)(cross_c[h.seg * NC + b].wordoff + h.segoff) + lane];
		};
		// NHI (a type that carries the number): the high-high hops this ITEM's slices carry (<= NH; SegItem::type)
		auto issue_data = [=](auto nhi, const SegHeads<GG>& h, const SegCrossWords<NC>& s, SegData<NC, decltype(nhi)::value, ROWS>& d) __attribute__((always_inline)) {
			constexpr int NHI = decltype(nhi)::value;
			uint32_t sg = 0;
#pragma unroll
			for (int b = 0; b < NC; b++) {
				const uint32_t sb = (uint32_t)cross_c[h.seg * NC + b].srcbase;
				if constexpr (ROWS == 1) { // a chain's row may exceed 4 GiB (L = 32: 6.0e8 positions): element index, 64-bit address
					d.xa[2 * b] = yrow0[sb + (s.x[b] & 0x1fffu)];
					if (!X1) d.xa[2 * b + 1] = yrow0[sb + ((s.x[b] >> 16) & 0x1fffu)];
				} else {
					const uint32_t at0 = (sb + (s.x[b] & 0x1fffu)) * 8u, at1 = (sb + ((s.x[b] >> 16) & 0x1fffu)) * 8u;
					d.xa[2 * b] = *(const double*)((const char*)yrow0 + at0);
					d.xb[2 * b] = *(const double*)((const char*)yrow1 + at0);
					d.xa[2 * b + 1] = *(const double*)((const char*)yrow0 + at1);
					d.xb[2 * b + 1] = *(const double*)((const char*)yrow1 + at1);
				}
				sg |= (((s.x[b] >> 14) & 3u) | ((s.x[b] >> 28) & 0xcu)) << (4 * b);
			}
			d.sg = sg;
			// High-high hops: position-preserving runs of the source segments.  A wave-level load costs a CU's vector-memory pipeline the same
			// ~17 cycles for 8 as for 16 bytes per lane (scripts/experiments/r04_tcp_rate.hip), so one load serves TWO hops: lanes 0..31 read
			// the positions (2l, 2l + 1) of hop 2b's run, lanes 32..63 those of hop 2b + 1 -- 16 bytes at an 8-byte aligned address.  The sums
			// change lanes once per slice (compute).  The last pair of a slice of `count` positions starts at count - 2 (-1 for a single
			// position: never in front of the row, a lone position is not the row's first).
			const int hst = min(2 * (lane & 31), h.count - 2);
#pragma unroll
			for (int b = 0; b < NHI; b += 2) {
				const SegHh* const en = hh_c + h.seg * NH + b + (lane >> 5);
				const int el = max(en->srcbase + (h.segoff + hst) * en->pad, 0);
				const SegPair8 t0 = *(const SegPair8*)(yrow0 + el);
				d.ha[b] = t0.x;
				d.ha[b + 1] = t0.y;
				if (ROWS == 2) {
					const SegPair8 t1 = *(const SegPair8*)(yrow1 + el);
					d.hb[b] = t1.x;
					d.hb[b + 1] = t1.y;
				}
			}
		};
		auto compute = [=, &dot](auto nhi, int jj, const SegHeads<GG>& h, const SegWinWords<GG>& s, const SegData<NC, decltype(nhi)::value, ROWS>& d) __attribute__((always_inline)) {
			constexpr int NHI = decltype(nhi)::value;
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
			for (int b = 0; b < 2 * NC; b += X1 ? 2 : 1) {
				// two bits per hop as a signed field: +1, -1 or 0 (no entry: the element read was the source segment's first)
				const double v = cross_c[h.seg * NC + (b >> 1)].val[b & 1] * (double)((int32_t)(d.sg << (30 - 2 * b)) >> 30);
				acc0 = fma(v, d.xa[b], acc0);
				if (ROWS == 2) acc1 = fma(v, d.xb[b], acc1);
			}
			{
				// this lane's half of the hops at the positions (hst, hst + 1); then position `lane` collects both halves' sums of its pair
				double p0x = 0.0, p0y = 0.0, p1x = 0.0, p1y = 0.0;
#pragma unroll
				for (int b = 0; b < NHI; b += 2) {
					const double v = hh_c[h.seg * NH + b + (lane >> 5)].val;
					p0x = fma(v, d.ha[b], p0x);
					p0y = fma(v, d.ha[b + 1], p0y);
					if (ROWS == 2) {
						p1x = fma(v, d.hb[b], p1x);
						p1y = fma(v, d.hb[b + 1], p1y);
					}
				}
				const int comp = (lane - min(lane & ~1, h.count - 2)) & 1; // where this position sits in its pair (the last pair starts at count - 2)
				const int src = 2 * (lane >> 1) + comp;
				// LDS instructions of a wave run in order, so the hardware needs no barrier; the wave barriers (no instruction) only keep the
				// compiler from moving the reads of other lanes' slots across this lane's store
				((double2*)tr_s)[lane] = double2 { p0x, p0y };
				__builtin_amdgcn_wave_barrier();
				acc0 += tr_s[src] + tr_s[64 + src];
				if (ROWS == 2) {
					__builtin_amdgcn_wave_barrier();
					((double2*)tr_s)[lane] = double2 { p1x, p1y };
					__builtin_amdgcn_wave_barrier();
					acc1 += tr_s[src] + tr_s[64 + src];
				}
			}
			const int lc = min(lane, h.count - 1);
			const int il = h.first + lc; // position in the item
			const double y0 = win_c[il + kSegWinPad], y1 = ROWS == 2 ? win_c[WS + il + kSegWinPad] : 0.0;
			acc0 = fma(dict_s[dcode_c[il + doff]], y0, acc0);
			if (ROWS == 2) acc1 = fma(dict_s[dcode_c[dstride + il + doff]], y1, acc1);
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
		auto run_slices = [&](auto nhi) __attribute__((always_inline)) {
			SegHeads<GG> h0, h1, h2, h3, h4;
			SegWinWords<GG> WA, WB;
			SegCrossWords<NC> XA, XB;
			SegData<NC, decltype(nhi)::value, ROWS> DA, DB;
			load_heads(wave, h0);
			load_heads(wave + NW, h1);
			load_heads(wave + 2 * NW, h2);
			load_cross_words(h0, XA);
			load_win_words(h0, WA);
			load_cross_words(h1, XB);
			issue_data(nhi, h0, XA, DA);
			load_cross_words(h2, XA);
			for (int jj = wave; jj < nsl; jj += 2 * NW) {
				issue_data(nhi, h1, XB, DB);
				load_win_words(h1, WB);
				load_heads(jj + 3 * NW, h3);
				load_cross_words(h3, XB);
				compute(nhi, jj, h0, WA, DA);
				issue_data(nhi, h2, XA, DA);
				load_win_words(h2, WA);
				load_heads(jj + 4 * NW, h4);
				load_cross_words(h4, XA);
				compute(nhi, jj + NW, h1, WB, DB);
				h0 = h2;
				h1 = h3;
				h2 = h4;
			}
		};
		// one block per workgroup (a chain: up to 12 high-high hops per segment, 6.0 on average): the loop for the 4, 8 or 12 this item holds
		if constexpr (ROWS == 1 && NH >= 12) {
			const int nhc = __builtin_amdgcn_readfirstlane(wk.I.type >> 16);
			if (nhc <= 4) run_slices(SegInt<4> {});
			else if (nhc <= 8) run_slices(SegInt<8> {});
			else if (NH == 12 || nhc <= 12) run_slices(SegInt<12> {});
			else run_slices(SegInt<NH> {}); // chains beyond 13 high sites (L = 32: 17 of them, up to 16 domain walls)
		} else
			run_slices(SegInt<NH> {});
		if (ROWS == 1) {
			if (has_next) stage_store(wn, cur ^ 1, S);
			__syncthreads(); // this item's slices are done with window `cur`; the next item's window is complete
			cur ^= 1;
			wk = wn;
		}
	}
	if (DOT) {
		const double r = block_sum_n<kSegThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

} // namespace lpp
