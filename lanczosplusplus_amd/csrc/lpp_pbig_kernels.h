// lpp_pbig_kernels.h -- the product-basis kernels (lpp_pb_kernels.h) for one-species spaces BEYOND one LDS window and vectors
// beyond 4 GiB: BASELINE config 5's sectors of the 4x5 lattice (N_up = 38760 ... 77520: a block's row is 0.3 - 0.6 MB, a vector
// 12 - 24 GB).  Same matrix form  H = 1 (x) T + C (x) 1 + D  (HubbardHelper.h:75-103 in the order of BasisHubbardLanczos.h:59-63),
// same pitched vectors, same split into an in-block kernel and a coupling kernel; what changes is how each keeps its gathers on chip.
//
//   k_pb_up_big      in-block part + diagonal.  A row no longer fits LDS, so it is cut into PIECES of W consecutive positions
//                    (W a multiple of 64; two 512-thread workgroups of ~80 KB share a CU, one stages while the other gathers).
//                    Basis words are ascending, i.e. the high sites are the major sort key: every hop among the low sites stays
//                    inside the piece and is an LDS gather exactly as in k_pb_up (16-bit window indices, value groups, edge-
//                    coloured slots).  The entries that LEAVE the piece (24-33 % at W = 8640 on the 4x5 lattice) are read from
//                    the row in memory: the pieces of one block are worked on by neighbouring workgroups of ONE XCD at the same
//                    time, so the row (0.6 MB) sits in that XCD's L2 while it is needed; two thirds of these entries move 64
//                    consecutive rows by the same amount (hops among the high sites) and share a slot = one coalesced 512-byte read.
//                    Every output row is formed once, from all its entries: nothing is read-modify-written (the round-2 kernel
//                    walked the row piece by piece and re-read / re-wrote x for every piece: 42 passes per product).
//   k_pb_down_parts  block couplings, panel-major as k_pb_down -- but a panel of 16 positions of ALL blocks is N_dn * 128 B = 5 MB
//                    at N_dn = 38760, more than one XCD's 4 MiB L2.  The source blocks are cut into NH ranges ("parts", ~1.2 MB of
//                    panel each); a workgroup keeps the sums of ALL its blocks for the current panel in registers (one double2 per
//                    task round) and walks the parts one after the other, every workgroup of the group within two parts of the
//                    others (bounded pacing), so that only ~3 parts of one panel are live in L2 at a time.  Addresses are 64-bit
//                    (the vector is 24 GB), formed from a 32-bit line number.
#pragma once
#include "lpp_pb_kernels.h"

namespace lpp {

constexpr int kPbMaxParts = 8;
constexpr int kPbBigThreads = 512;
// the coupling kernel keeps one double2 per task round and lane for the whole panel: 512 threads (8 waves, up to 256 registers
// each) per CU rather than 1024 with 128 -- ten rounds of sums, two chunks of gathers and the addresses spill at 128
constexpr int kPbPartsThreads = 512;
#ifndef LPP_PB_PARTS_DEPTH
#define LPP_PB_PARTS_DEPTH 4
#endif
constexpr int kPbPartsDepth = LPP_PB_PARTS_DEPTH; // pairs of gathers in flight per wave

struct PbUpBigArgs {
	// in-window entries: for slice j (of the ROW) and value group g, tw_len[j*G+g] chunks from chunk tw_off[j*G+g] (as PbUpArgs)
	const uint32_t* tw;
	const int32_t* tw_off;
	const uint16_t* tw_len;
	// entries that leave the window: f_len[j] slots of 64 words from slot f_off[j]; word = position in the row (24 bits) | group << 24
	const uint32_t* fw;
	const int32_t* f_off;
	const uint16_t* f_len;
	int G;
	double gval[kPbMaxGroups + 1]; // gval[G] = 0.0 (filling words)
	const double* dict; // 256 doubles (diagonal codes)
	const uint8_t* dcode; // one code per row, pitched like the vectors
	int64_t n_up, pitch, n_blk;
	int W, npieces; // window positions, pieces per row
	const double* y;
	double* u; // out: alpha (T y + D y), pitched
	double* partial; // per-workgroup Re<y|u> (null: not wanted)
	EpiScale sc; // only alpha is used
};

// LDS: window (W + 32 zero slots) | dcode[W] | dict[256] | gval[9] | smem[THREADS/64]
__host__ __device__ inline size_t pb_big_lds_bytes(int W)
{
	return ((sizeof(double) * (size_t)(W + kPbZeroSlots) + (size_t)W + 15) & ~(size_t)15) + sizeof(double) * (256 + kPbMaxGroups + 1 + kPbBigThreads / 64) + 16;
}

// Per slice a wave needs: the list heads (scalars), the template words of every value group and the far words (vector loads
// that depend on the heads), the far elements (loads that depend on the far words) and the LDS gathers (which depend on the
// template words).  Run one after the other that is four dependent round trips per slice and 8 + 8 waves per CU cannot hide them
// (first version: 82 ms per product at 3.0e9 states, 88 % of the wave cycles waiting).  So, as in k_pb_up, everything is
// requested ahead into fixed register sets: heads two slices ahead, words one slice ahead, and within a slice the far elements
// are requested first and consumed last, behind the LDS gathers.
constexpr int kBigPre = 4; // template chunks of every (slice, group) requested one slice ahead
constexpr int kBigFarPre = 8; // far slots requested one slice ahead

template <int GG> struct BigHeads { // wave-uniform
	int nc[GG], off[GG], nf, foff;
};
template <int GG, int PRE = kBigPre> struct BigWords {
	uint2 w[GG][PRE];
	uint32_t f[kBigFarPre];
};

// GT = number of value groups (1, 2 or 4: unrolled with look-ahead; 0: any G <= 8, plain loop).  Four groups is what complex hoppings
// realified over (re, im) pairs come to (+-cos, +-sin of a Peierls phase; +-t and +-lambda of a Kane-Mele model): their lists are half as
// long as a real matrix's two, so three chunks per group are requested ahead (PRE; 118 registers).
template <bool DOT, int GT, int PRE = kBigPre> __global__ __launch_bounds__(kPbBigThreads, 4) void k_pb_up_big(PbUpBigArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // at LDS address 0: a 16-bit template entry * 8 IS the byte address (pb_lds_abs)
	uint8_t* dcode_s = (uint8_t*)(win + a.W + kPbZeroSlots);
	double* dict_s = (double*)(lds_raw + ((sizeof(double) * (size_t)(a.W + kPbZeroSlots) + (size_t)a.W + 15) & ~(size_t)15));
	double* gv_s = dict_s + 256;
	double* smem = gv_s + kPbMaxGroups + 1;
	for (int i = threadIdx.x; i < 256; i += kPbBigThreads) dict_s[i] = a.dict[i];
	if (threadIdx.x <= kPbMaxGroups) gv_s[threadIdx.x] = threadIdx.x <= (unsigned)a.G ? a.gval[threadIdx.x] : 0.0;
	if (threadIdx.x < kPbZeroSlots) win[a.W + threadIdx.x] = 0.0;
	double alpha, beta_unused;
	epi_coeffs(a.sc, alpha, beta_unused);
	constexpr int NW = kPbBigThreads / 64;
	constexpr int GG = GT > 0 ? GT : 1;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint2* const tw2 = (const uint2*)a.tw;
	const uint32_t* const fw = a.fw;
	const int32_t* const tw_off = a.tw_off;
	const uint16_t* const tw_len = a.tw_len;
	const int32_t* const f_off = a.f_off;
	const uint16_t* const f_len = a.f_len;
	const int G = a.G, W = a.W, npieces = a.npieces, n_up = (int)a.n_up;
	double* const uout = a.u;
	double gv[GG];
#pragma unroll
	for (int g = 0; g < GG; g++) gv[g] = a.gval[g];
	double dot = 0.0;
	// The pieces of one block go to neighbouring workgroups of one XCD (workgroups w and w + 8 share an XCD under round-robin
	// dispatch -- for speed only): the XCD's workgroups walk  seq = (block, piece)  together, so the ~nslots / npieces rows in
	// flight stay in that L2 for the reads that leave a window.
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1;
	const int xcd = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int64_t slot = nx == 8 ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x, nslots = gridDim.x / nx;
	const int64_t nbx = (a.n_blk - xcd + nx - 1) / nx; // blocks of this XCD: xcd, xcd + nx, ...
	auto gather4 = [=](const uint2& w, double& s0, double& s1) __attribute__((always_inline)) {
		s0 += pb_lds_abs(pb_lo8(w.x));
		s1 += pb_lds_abs(pb_hi8(w.x));
		s0 += pb_lds_abs(pb_lo8(w.y));
		s1 += pb_lds_abs(pb_hi8(w.y));
	};
	for (int64_t seq = slot; seq < nbx * npieces; seq += nslots) {
		const int64_t bl = seq / npieces;
		const int q = (int)(seq - bl * npieces);
		const int64_t blk = bl * nx + xcd;
		const int c0 = q * W; // first position of the piece
		const int wlen = min(W, (int)a.pitch - c0); // staged positions (the padding behind the row holds zeros), a multiple of 16
		const int64_t rowbase = blk * a.pitch;
		const double* const yrow = a.y + rowbase;
		const double2* yb = (const double2*)(yrow + c0);
		const uint4* db = (const uint4*)(a.dcode + rowbase + c0);
		__syncthreads(); // everyone is done with the previous window
		{
			constexpr int NS = 4; // 16-byte loads per thread in flight
			const int p2 = wlen >> 1;
			for (int i0 = threadIdx.x; i0 < p2; i0 += NS * kPbBigThreads) {
				double2 t[NS];
				int idx[NS];
#pragma unroll
				for (int k = 0; k < NS; k++) idx[k] = min(i0 + k * kPbBigThreads, p2 - 1);
#pragma unroll
				for (int k = 0; k < NS; k++) t[k] = yb[idx[k]];
#pragma unroll
				for (int k = 0; k < NS; k++) ((double2*)win)[idx[k]] = t[k]; // clamped lanes re-store the last pair
			}
			for (int i0 = threadIdx.x; i0 < (wlen >> 4); i0 += kPbBigThreads) ((uint4*)dcode_s)[i0] = db[i0];
		}
		__syncthreads();
		const int j0 = c0 >> 6, nsl = (min(c0 + W, n_up) - c0 + 63) >> 6; // slices of this piece
		auto epilogue = [=, &dot](int jj, double acc) __attribute__((always_inline)) {
			const int il_raw = jj * 64 + lane; // position inside the window
			const bool valid = c0 + il_raw < n_up;
			const int il = valid ? il_raw : n_up - 1 - c0;
			const double yc = win[il];
			acc = fma(dict_s[dcode_s[il]], yc, acc);
			if (valid) {
				const double uv = alpha * acc;
				__builtin_nontemporal_store(uv, &uout[rowbase + c0 + il]);
				if (DOT) dot += yc * uv;
			}
		};
		if (GT > 0) {
			auto load_heads = [=](int jj, BigHeads<GG>& h) __attribute__((always_inline)) {
				const int j = __builtin_amdgcn_readfirstlane(j0 + min(jj, nsl - 1)); // beyond the piece: a valid slice, never used
#pragma unroll
				for (int g = 0; g < GG; g++) {
					h.nc[g] = tw_len[j * GG + g];
					h.off[g] = tw_off[j * GG + g];
				}
				h.nf = f_len[j];
				h.foff = f_off[j];
			};
			auto load_words = [=](const BigHeads<GG>& h, BigWords<GG, PRE>& s) __attribute__((always_inline)) {
#pragma unroll
				for (int g = 0; g < GG; g++) {
					const uint2* wp = tw2 + (size_t)h.off[g] * 64 + lane;
#pragma unroll
					for (int c = 0; c < PRE; c++) s.w[g][c] = wp[c * 64]; // chunks beyond the list belong to the next list (or the slack): never used
				}
				const uint32_t* fp = fw + (size_t)h.foff * 64 + lane;
#pragma unroll
				for (int k = 0; k < kBigFarPre; k++) s.f[k] = fp[k * 64];
			};
			auto compute = [=](int jj, const BigHeads<GG>& h, const BigWords<GG, PRE>& s) __attribute__((always_inline)) {
				if (jj >= nsl) return; // wave-uniform
				// far elements first (requested here, used at the end); slots beyond the list are not requested
				double fv[kBigFarPre];
				const int nf = h.nf;
#pragma unroll
				for (int k = 0; k < kBigFarPre; k += 4)
					if (k < nf) { // lists are whole groups of 4 slots
#pragma unroll
						for (int t = 0; t < 4; t++) fv[k + t] = yrow[s.f[k + t] & 0xffffffu];
					}
				double acc = 0.0;
#pragma unroll
				for (int g = 0; g < GG; g++) {
					const int nc = h.nc[g];
					double s0 = 0.0, s1 = 0.0;
					if (nc >= 2) {
						gather4(s.w[g][0], s0, s1);
						gather4(s.w[g][1], s0, s1);
					} else if (nc == 1) {
						gather4(s.w[g][0], s0, s1);
					}
					if constexpr (PRE >= 4) {
						if (nc >= 4) {
							gather4(s.w[g][2], s0, s1);
							gather4(s.w[g][3], s0, s1);
						} else if (nc == 3) {
							gather4(s.w[g][2], s0, s1);
						}
					} else if constexpr (PRE == 3) {
						if (nc >= 3) gather4(s.w[g][2], s0, s1);
					}
					if (nc > PRE) {
						const uint2* wp = tw2 + (size_t)h.off[g] * 64 + lane;
						for (int c = PRE; c < nc; c++) {
							const uint2 wr = wp[c * 64];
							gather4(wr, s0, s1);
						}
					}
					acc = fma(gv[g], s0 + s1, acc);
				}
				if (nf > kBigFarPre) { // longer far lists: the rest streamed
					const uint32_t* fp = fw + (size_t)h.foff * 64 + lane;
					for (int k = kBigFarPre; k < nf; k += 4) {
						uint32_t w[4];
#pragma unroll
						for (int t = 0; t < 4; t++) w[t] = fp[(k + t) * 64];
						double v[4];
#pragma unroll
						for (int t = 0; t < 4; t++) v[t] = yrow[w[t] & 0xffffffu];
#pragma unroll
						for (int t = 0; t < 4; t++) acc = fma(gv_s[w[t] >> 24], v[t], acc);
					}
				}
#pragma unroll
				for (int k = 0; k < kBigFarPre; k += 4)
					if (k < nf) {
#pragma unroll
						for (int t = 0; t < 4; t++) acc = fma(gv_s[s.f[k + t] >> 24], fv[k + t], acc);
					}
				epilogue(jj, acc);
			};
			BigHeads<GG> h0, h1, h2, h3;
			BigWords<GG, PRE> wa, wb;
			load_heads(wave, h0);
			load_heads(wave + NW, h1);
			load_words(h0, wa);
			for (int jj = wave; jj < nsl; jj += 2 * NW) {
				load_heads(jj + 2 * NW, h2);
				load_words(h1, wb);
				compute(jj, h0, wa);
				load_heads(jj + 3 * NW, h3);
				load_words(h2, wa);
				compute(jj + NW, h1, wb);
				h0 = h2;
				h1 = h3;
			}
		} else {
			for (int jj = wave; jj < nsl; jj += NW) {
				const int j = __builtin_amdgcn_readfirstlane(j0 + jj); // wave-uniform: the list heads below are scalar loads
				double acc = 0.0;
				{
					const int nf = f_len[j];
					const uint32_t* fp = fw + (size_t)f_off[j] * 64 + lane;
					for (int s = 0; s < nf; s += 4) { // lists are padded to whole groups of 4 slots
						uint32_t w[4];
#pragma unroll
						for (int k = 0; k < 4; k++) w[k] = fp[(s + k) * 64];
						double v[4];
#pragma unroll
						for (int k = 0; k < 4; k++) v[k] = yrow[w[k] & 0xffffffu];
#pragma unroll
						for (int k = 0; k < 4; k++) acc = fma(gv_s[w[k] >> 24], v[k], acc);
					}
				}
				for (int g = 0; g < G; g++) { // wave-uniform trip counts
					const int nc = tw_len[j * G + g];
					const uint2* wp = tw2 + (size_t)tw_off[j * G + g] * 64 + lane;
					double s0 = 0.0, s1 = 0.0;
					for (int c = 0; c < nc; c++) {
						const uint2 wr = wp[c * 64];
						gather4(wr, s0, s1);
					}
					acc = fma(gv_s[g], s0 + s1, acc);
				}
				epilogue(jj, acc);
			}
		}
	}
	if (DOT) {
		const double r = block_sum_n<kPbBigThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// The same with TWO blocks per template read.  The template of a long row is large (80 bytes per position: 6.2 MB at N_up = 77520)
// and every (block, piece) streams its share of it: 240 GB of the 374 GB k_pb_up_big reads per product at 3.0e9 states are template
// words, which cannot stay in L2 beside the rows.  Here a 1024-thread workgroup stages the same piece of two blocks (two windows; the
// second one within the 64 KB reach of an LDS instruction's offset field, so a template entry still is the address) and applies every
// word it loads to both: half the template traffic and half the word decoding per output row.
// ---------------------------------------------------------------------------------------------
constexpr int kPbBig2Threads = 1024;
constexpr int kBig2FarPre = 4;

__host__ __device__ inline size_t pb_big2_lds_bytes(int W)
{
	return ((2 * sizeof(double) * (size_t)(W + kPbZeroSlots) + 2 * (size_t)W + 15) & ~(size_t)15) + sizeof(double) * (256 + kPbMaxGroups + 1 + kPbBig2Threads / 64) + 16;
}

template <int GG> struct Big2Words {
	uint2 w[GG][kPbPreMax]; // only the first depth(g) chunks of group g are touched
	uint32_t f[kBig2FarPre];
};

// PRE0: look-ahead chunks of value group 0 (group 1: 2 kBigPre - PRE0), as in k_pb_up; the words of a chunk are requested only if the
// list holds it (at the (7,6) sector of the 4x5 lattice the lists are 2 and 3-4 chunks long: 64 bytes per row were requested for 45).
// Four value groups (complex hoppings realified, as in k_pb_up_big<.., 4, 3>): PRE0 chunks of EVERY group.
template <bool DOT, int GT, int PRE0 = kBigPre> __global__ __launch_bounds__(kPbBig2Threads) void k_pb_up_big2(PbUpBigArgs a)
{
	static_assert(GT == 1 || GT == 2 || GT == 4, "unrolled value groups only");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // window of block 0 at LDS address 0, of block 1 at (W + 32) * 8
	const int WS = a.W + kPbZeroSlots; // window stride in elements
	uint8_t* dcode_s = (uint8_t*)(win + 2 * WS); // [2][W]
	double* dict_s = (double*)(lds_raw + ((2 * sizeof(double) * (size_t)WS + 2 * (size_t)a.W + 15) & ~(size_t)15));
	double* gv_s = dict_s + 256;
	double* smem = gv_s + kPbMaxGroups + 1;
	for (int i = threadIdx.x; i < 256; i += kPbBig2Threads) dict_s[i] = a.dict[i];
	if (threadIdx.x <= kPbMaxGroups) gv_s[threadIdx.x] = threadIdx.x <= (unsigned)a.G ? a.gval[threadIdx.x] : 0.0;
	if (threadIdx.x < 2 * kPbZeroSlots) win[(threadIdx.x >> 5) * WS + a.W + (threadIdx.x & 31)] = 0.0;
	double alpha, beta_unused;
	epi_coeffs(a.sc, alpha, beta_unused);
	constexpr int NW = kPbBig2Threads / 64;
	constexpr int GG = GT;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint2* const tw2 = (const uint2*)a.tw;
	const uint32_t* const fw = a.fw;
	const int32_t* const tw_off = a.tw_off;
	const uint16_t* const tw_len = a.tw_len;
	const int32_t* const f_off = a.f_off;
	const uint16_t* const f_len = a.f_len;
	const int W = a.W, npieces = a.npieces, n_up = (int)a.n_up;
	const uint32_t wbytes = (uint32_t)WS * 8u; // byte distance of the two windows (< 65536: pb_build)
	double* const uout = a.u;
	double gv[GG];
#pragma unroll
	for (int g = 0; g < GG; g++) gv[g] = a.gval[g];
	double dot = 0.0;
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1;
	const int xcd = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int64_t slot = nx == 8 ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x, nslots = gridDim.x / nx;
	const int64_t nbx = (a.n_blk - xcd + nx - 1) / nx; // blocks of this XCD: xcd, xcd + nx, ...
	const int64_t npairs = (nbx + 1) >> 1;
	auto gather4x2 = [=](const uint2& w, double& a0, double& a1, double& b0, double& b1) __attribute__((always_inline)) {
		const uint32_t p0 = pb_lo8(w.x), p1 = pb_hi8(w.x), p2 = pb_lo8(w.y), p3 = pb_hi8(w.y);
		a0 += pb_lds_abs(p0);
		b0 += pb_lds_abs(p0 + wbytes);
		a1 += pb_lds_abs(p1);
		b1 += pb_lds_abs(p1 + wbytes);
		a0 += pb_lds_abs(p2);
		b0 += pb_lds_abs(p2 + wbytes);
		a1 += pb_lds_abs(p3);
		b1 += pb_lds_abs(p3 + wbytes);
	};
	for (int64_t seq = slot; seq < npairs * npieces; seq += nslots) {
		const int64_t pr = seq / npieces;
		const int q = (int)(seq - pr * npieces);
		const int64_t blk0 = (2 * pr) * nx + xcd;
		const bool two = 2 * pr + 1 < nbx; // the last pair of an odd count holds one block: its twin re-reads it and stores nothing
		const int64_t blk1 = two ? (2 * pr + 1) * nx + xcd : blk0;
		const int c0 = q * W;
		const int wlen = min(W, (int)a.pitch - c0);
		const int64_t rowbase0 = blk0 * a.pitch, rowbase1 = blk1 * a.pitch;
		const double* const yrow0 = a.y + rowbase0;
		const double* const yrow1 = a.y + rowbase1;
		__syncthreads(); // everyone is done with the previous windows
		{
			constexpr int NS = 4;
			const int p2 = wlen >> 1;
			for (int i0 = threadIdx.x; i0 < 2 * p2; i0 += NS * kPbBig2Threads) { // [0, p2): block 0, [p2, 2 p2): block 1
				double2 t[NS];
				int idx[NS];
#pragma unroll
				for (int k = 0; k < NS; k++) idx[k] = min(i0 + k * kPbBig2Threads, 2 * p2 - 1);
#pragma unroll
				for (int k = 0; k < NS; k++) {
					const bool second = idx[k] >= p2;
					t[k] = ((const double2*)((second ? yrow1 : yrow0) + c0))[second ? idx[k] - p2 : idx[k]];
				}
#pragma unroll
				for (int k = 0; k < NS; k++) {
					const bool second = idx[k] >= p2;
					((double2*)(win + (second ? WS : 0)))[second ? idx[k] - p2 : idx[k]] = t[k];
				}
			}
			const int p16 = wlen >> 4;
			for (int i0 = threadIdx.x; i0 < 2 * p16; i0 += kPbBig2Threads) {
				const bool second = i0 >= p16;
				const int k = second ? i0 - p16 : i0;
				((uint4*)(dcode_s + (second ? W : 0)))[k] = ((const uint4*)(a.dcode + (second ? rowbase1 : rowbase0) + c0))[k];
			}
		}
		__syncthreads();
		const int j0 = c0 >> 6, nsl = (min(c0 + W, n_up) - c0 + 63) >> 6;
		auto epilogue = [=, &dot](int jj, double acc0, double acc1) __attribute__((always_inline)) {
			const int il_raw = jj * 64 + lane;
			const bool valid = c0 + il_raw < n_up;
			const int il = valid ? il_raw : n_up - 1 - c0;
			const double y0 = win[il], y1 = win[WS + il];
			acc0 = fma(dict_s[dcode_s[il]], y0, acc0);
			acc1 = fma(dict_s[dcode_s[W + il]], y1, acc1);
			if (valid) {
				const double u0 = alpha * acc0, u1 = alpha * acc1;
				__builtin_nontemporal_store(u0, &uout[rowbase0 + c0 + il]);
				if (DOT) dot += y0 * u0;
				if (two) {
					__builtin_nontemporal_store(u1, &uout[rowbase1 + c0 + il]);
					if (DOT) dot += y1 * u1;
				}
			}
		};
		auto load_heads = [=](int jj, BigHeads<GG>& h) __attribute__((always_inline)) {
			const int j = __builtin_amdgcn_readfirstlane(j0 + min(jj, nsl - 1));
#pragma unroll
			for (int g = 0; g < GG; g++) {
				h.nc[g] = tw_len[j * GG + g];
				h.off[g] = tw_off[j * GG + g];
			}
			h.nf = f_len[j];
			h.foff = f_off[j];
		};
		auto load_words = [=](const BigHeads<GG>& h, Big2Words<GG>& s) __attribute__((always_inline)) {
#pragma unroll
			for (int g = 0; g < GG; g++) {
				const uint2* wp = tw2 + (size_t)h.off[g] * 64 + lane;
				const int depth = GG == 2 ? (g == 0 ? PRE0 : 2 * kBigPre - PRE0) : GG == 4 ? PRE0 : kBigPre;
				const int nc = __builtin_amdgcn_readfirstlane(h.nc[g]); // scalar branches
#pragma unroll
				for (int c = 0; c < kPbPreMax; c++)
					if (c < depth && (c < 1 || c < nc)) s.w[g][c] = wp[c * 64];
			}
			const uint32_t* fp = fw + (size_t)h.foff * 64 + lane;
#pragma unroll
			for (int k = 0; k < kBig2FarPre; k++) s.f[k] = fp[k * 64];
		};
		auto compute = [=](int jj, const BigHeads<GG>& h, const Big2Words<GG>& s) __attribute__((always_inline)) {
			if (jj >= nsl) return; // wave-uniform
			const int nf = h.nf;
			double fv0[kBig2FarPre], fv1[kBig2FarPre];
			if (nf > 0) { // lists are whole groups of 4 slots
#pragma unroll
				for (int t = 0; t < kBig2FarPre; t++) {
					const uint32_t c = s.f[t] & 0xffffffu;
					fv0[t] = yrow0[c];
					fv1[t] = yrow1[c];
				}
			}
			double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
			for (int g = 0; g < GG; g++) {
				const int nc = __builtin_amdgcn_readfirstlane(h.nc[g]);
				const int depth = GG == 2 ? (g == 0 ? PRE0 : 2 * kBigPre - PRE0) : GG == 4 ? PRE0 : kBigPre;
				double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
				for (int c = 0; c < kPbPreMax; c += 2) { // pairs of chunks
					if (c + 1 < depth) {
						if (nc >= c + 2) {
							gather4x2(s.w[g][c], a0, a1, b0, b1);
							gather4x2(s.w[g][c + 1], a0, a1, b0, b1);
						} else if (nc == c + 1) {
							gather4x2(s.w[g][c], a0, a1, b0, b1);
						}
					} else if (c < depth) {
						if (nc >= c + 1) gather4x2(s.w[g][c], a0, a1, b0, b1);
					}
				}
				if (nc > depth) {
					const uint2* wp = tw2 + (size_t)h.off[g] * 64 + lane;
					for (int c = depth; c < nc; c++) {
						const uint2 wr = wp[c * 64];
						gather4x2(wr, a0, a1, b0, b1);
					}
				}
				acc0 = fma(gv[g], a0 + a1, acc0);
				acc1 = fma(gv[g], b0 + b1, acc1);
			}
			if (nf > kBig2FarPre) { // the rest of the far list, streamed
				const uint32_t* fp = fw + (size_t)h.foff * 64 + lane;
				for (int k = kBig2FarPre; k < nf; k += 4) {
					uint32_t w[4];
#pragma unroll
					for (int t = 0; t < 4; t++) w[t] = fp[(k + t) * 64];
					double v0[4], v1[4];
#pragma unroll
					for (int t = 0; t < 4; t++) {
						v0[t] = yrow0[w[t] & 0xffffffu];
						v1[t] = yrow1[w[t] & 0xffffffu];
					}
#pragma unroll
					for (int t = 0; t < 4; t++) {
						const double gq = gv_s[w[t] >> 24];
						acc0 = fma(gq, v0[t], acc0);
						acc1 = fma(gq, v1[t], acc1);
					}
				}
			}
			if (nf > 0) {
#pragma unroll
				for (int t = 0; t < kBig2FarPre; t++) {
					const double gq = gv_s[s.f[t] >> 24];
					acc0 = fma(gq, fv0[t], acc0);
					acc1 = fma(gq, fv1[t], acc1);
				}
			}
			epilogue(jj, acc0, acc1);
		};
		BigHeads<GG> h0, h1, h2, h3;
		Big2Words<GG> wa, wb;
		load_heads(wave, h0);
		load_heads(wave + NW, h1);
		load_words(h0, wa);
		for (int jj = wave; jj < nsl; jj += 2 * NW) {
			load_heads(jj + 2 * NW, h2);
			load_words(h1, wb);
			compute(jj, h0, wa);
			load_heads(jj + 3 * NW, h3);
			load_words(h2, wa);
			compute(jj + NW, h1, wb);
			h0 = h2;
			h1 = h3;
		}
	}
	if (DOT) {
		const double r = block_sum_n<kPbBig2Threads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// block couplings over parts of the source range:  z[b][i] = alpha sum_k C[b][b'_k] y[b'_k][i]   (+ Re<y|z> partial)
// ---------------------------------------------------------------------------------------------
struct PbDownPartsArgs {
	int64_t pitch, n_blk;
	int npanels; // pitch / 16
	int ids_per_wg; // blocks owned by one workgroup
	int nparts; // NH
	int ent_cap; // list entries a workgroup holds at most (LDS sizing)
	const int64_t* c_ptr; // couplings: CSR over blocks, off-diagonal, ascending
	const int32_t* c_col;
	const uint8_t* c_code;
	const int32_t* c_pstart; // [n_blk][nparts + 1] first entry of part h within the block's list (part h = a range of source blocks)
	const int32_t* order; // blocks of every workgroup's range by decreasing list length
	const double* dict;
	const double* y;
	double* z;
	double* partial; // per-workgroup Re<y|z> (null: not wanted)
	EpiScale sc; // only alpha is used
	int* pace; // [8][panels of a group * nparts] finished-workgroup counters (zeroed before the launch); null: free-running
	int pace_stride; // counters per group
};

// LDS image of a workgroup's coupling lists, compact (a padded image -- every part of every list as long as the longest --
// is 170 KB at N_dn = 38760 in four parts; this one 82 KB): line[ids] | pstart[ids][NH+1] (places) | n4[tasks][NH] | idx[ent] | code[ent]
__host__ __device__ inline size_t pb_parts_lds_bytes(int ids_per_wg, int ent_cap, int nparts)
{
	const size_t ngroups = (size_t)(ids_per_wg + 7) / 8;
	return (size_t)ids_per_wg * (4 + 2 * ((size_t)nparts + 1)) + ((ngroups * (size_t)nparts + 1) & ~(size_t)1) + (size_t)(ent_cap + 8) * 3 + 64;
}

template <int THREADS, int MAXR> __global__ __launch_bounds__(THREADS) void k_pb_down_parts(PbDownPartsArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	__shared__ double dict_s[256];
	__shared__ double smem_d[THREADS / 64];
	const int NH = a.nparts;
	const int ngmax = (a.ids_per_wg + 7) >> 3;
	uint32_t* line_s = (uint32_t*)lds_raw; // [ids] first 128-byte line of the block's row
	uint16_t* ps_s = (uint16_t*)(line_s + a.ids_per_wg); // [ids][NH+1] first place of part h of the block's list
	uint8_t* n4_s = (uint8_t*)(ps_s + (size_t)a.ids_per_wg * (NH + 1)); // [ngmax][NH] trip count (pairs of gathers) of task g in part h
	uint16_t* idx_s = (uint16_t*)(n4_s + (((size_t)ngmax * NH + 1) & ~(size_t)1)); // [ent] source blocks
	uint8_t* code_s = (uint8_t*)(idx_s + a.ent_cap + 8); // [ent]
	for (int i = threadIdx.x; i < 256; i += THREADS) dict_s[i] = a.dict[i];
	double alpha, beta_unused;
	epi_coeffs(a.sc, alpha, beta_unused);
	constexpr int NW = THREADS / 64;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 3, c = lane & 7;
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1;
	const int grp = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int slot = nx == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
	const int nslots = (int)(gridDim.x / nx);
	const int64_t b0 = (int64_t)slot * a.ids_per_wg;
	const int nown = (int)max((int64_t)0, min((int64_t)a.ids_per_wg, a.n_blk - b0));
	const uint32_t lines_per_row = (uint32_t)(a.pitch >> 4);
	// places: the lists of the workgroup's blocks one after the other (in `order`); wave 0 scans the lengths
	if (wave == 0) {
		int run = 0;
		for (int i0 = 0; i0 < nown; i0 += 64) {
			const int il = i0 + lane;
			const int64_t b = il < nown ? (int64_t)a.order[b0 + il] : 0;
			const int len = il < nown ? (int)(a.c_ptr[b + 1] - a.c_ptr[b]) : 0;
			int incl = len; // inclusive scan over the wave
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const int t = __shfl_up(incl, off, 64);
				if (lane >= off) incl += t;
			}
			if (il < nown) {
				const int first = run + incl - len;
				line_s[il] = (uint32_t)b * lines_per_row;
				for (int h = 0; h <= NH; h++) ps_s[il * (NH + 1) + h] = (uint16_t)(first + a.c_pstart[b * (NH + 1) + h]);
			}
			run += __shfl(incl, 63, 64);
		}
	}
	__syncthreads();
	for (int il = wave; il < nown; il += NW) { // one wave per list: lists are <= 64 long in practice, longer ones loop
		const int64_t b = a.order[b0 + il];
		const int64_t p0 = a.c_ptr[b];
		const int len = (int)(a.c_ptr[b + 1] - p0), first = ps_s[il * (NH + 1)];
		for (int k = lane; k < len; k += 64) {
			idx_s[first + k] = (uint16_t)a.c_col[p0 + k];
			code_s[first + k] = a.c_code[p0 + k];
		}
	}
	const int ngroups = (nown + 7) >> 3;
	for (int i = threadIdx.x; i < ngmax * NH; i += THREADS) {
		const int g = i / NH, h = i - g * NH;
		int mx = 0;
		for (int t = 0; t < 8; t++)
			if (g * 8 + t < nown) mx = max(mx, (int)ps_s[(g * 8 + t) * (NH + 1) + h + 1] - (int)ps_s[(g * 8 + t) * (NH + 1) + h]);
		n4_s[i] = (uint8_t)((mx + 1) >> 1);
	}
	__syncthreads();
	double dot = 0.0;
	const char* ysrc = (const char*)a.y;
	int phase = 0; // counts (panel, part) pairs of this group
	for (int p = grp; p < a.npanels; p += nx) {
		const uint32_t colb = (uint32_t)(c * 16); // byte offset of this lane's two positions inside a panel line
		double2 acc[MAXR];
#pragma unroll
		for (int r = 0; r < MAXR; r++) acc[r] = double2 { 0.0, 0.0 };
		for (int h = 0; h < NH; h++, phase++) {
			if (a.pace && phase >= 2) {
				// bounded wait: the part before the previous one must be finished by every workgroup of the group
				if (threadIdx.x == 0) {
					const int* cnt = a.pace + (int64_t)grp * a.pace_stride + (phase - 2);
					for (int spin = 0; spin < 8192 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nslots; spin++)
						__builtin_amdgcn_s_sleep(8);
				}
				__syncthreads();
			}
#pragma unroll
			for (int r = 0; r < MAXR; r++) {
				const int g = wave + r * NW;
				if (g < ngroups) { // wave-uniform
					const int il = min(g * 8 + sub, nown - 1);
					const int n2 = __builtin_amdgcn_readfirstlane((int)n4_s[g * NH + h]); // trip count in pairs of gathers
					const int first = ps_s[il * (NH + 1) + h], len = (int)ps_s[il * (NH + 1) + h + 1] - first;
					const uint32_t own_line = line_s[il] + (uint32_t)p;
					// four pairs of gathers in flight, one fixed buffer per stage (a rotated buffer would wait for the loads it holds).
					// Places beyond this block's list: its own line (read at the end of the panel anyway) times +0.0
					double2 gq[kPbPartsDepth][2];
					auto issue = [&](int ch, double2* gbuf) __attribute__((always_inline)) {
#pragma unroll
						for (int qq = 0; qq < 2; qq++) {
							const int k = ch * 2 + qq;
							const uint32_t ln = k < len ? (uint32_t)idx_s[first + k] * lines_per_row + (uint32_t)p : own_line;
							gbuf[qq] = *(const double2*)(ysrc + (((uint64_t)ln << 7) + colb));
						}
					};
					auto consume = [&](int ch, const double2* gbuf) __attribute__((always_inline)) {
#pragma unroll
						for (int qq = 0; qq < 2; qq++) {
							const int k = ch * 2 + qq;
							const double v = k < len ? dict_s[code_s[first + k]] : 0.0;
							acc[r].x = fma(v, gbuf[qq].x, acc[r].x);
							acc[r].y = fma(v, gbuf[qq].y, acc[r].y);
						}
					};
					// software pipeline of depth D over pairs: stage t of iteration ch uses buffer t (the loop is unrolled by D, so
					// every buffer index is a constant)
					constexpr int D = kPbPartsDepth;
#pragma unroll
					for (int t = 0; t < D - 1; t++)
						if (t < n2) issue(t, gq[t]);
					for (int ch = 0; ch < n2; ch += D) { // wave-uniform conditions
#pragma unroll
						for (int t = 0; t < D; t++) {
							if (ch + t < n2) {
								if (ch + t + D - 1 < n2) issue(ch + t + D - 1, gq[(t + D - 1) % D]);
								consume(ch + t, gq[t]);
							}
						}
					}
				}
			}
			if (a.pace) {
				__syncthreads();
				if (threadIdx.x == 0) __hip_atomic_fetch_add(a.pace + (int64_t)grp * a.pace_stride + phase, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
#pragma unroll
		for (int r = 0; r < MAXR; r++) {
			const int g = wave + r * NW;
			if (g < ngroups) {
				const int il = min(g * 8 + sub, nown - 1);
				const bool valid = g * 8 + sub < nown;
				const uint64_t off = ((uint64_t)(line_s[il] + (uint32_t)p) << 7) + colb;
				const double2 yown = *(const double2*)(ysrc + off); // the panel is in L2
				if (valid) {
					double2* const zp = (double2*)((char*)a.z + off);
					const double zx = alpha * acc[r].x, zy = alpha * acc[r].y;
					__builtin_nontemporal_store(zx, &zp->x);
					__builtin_nontemporal_store(zy, &zp->y);
					dot += yown.x * zx + yown.y * zy;
				}
			}
		}
	}
	if (a.partial) {
		const double r = block_sum_n<THREADS / 64>(dot, smem_d);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

} // namespace lpp
