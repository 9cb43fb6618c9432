// lpp_pb_kernels.h -- stored matrices of PRODUCT-BASIS form, the structure of HubbardHelper::setupHamiltonian's matrix
// (reference src/Models/HubbardOneOrbital/HubbardHelper.h:75-103) in the BasisHubbardLanczos order index = i_up + i_down*N_up
// (BasisHubbardLanczos.h:59-63):
//     H = 1 (x) T  +  C (x) 1  +  D,        row = (block b = i_down, position i = i_up)
//   T  in-block matrix, the same in every block (up-hops; off-diagonal part),          N_up x N_up,  ~17 entries per row
//   C  block couplings: entry (b, b') couples position i of block b to position i of b' (down-hops; off-diagonal), N_dn x N_dn
//   D  the diagonal, one value per row (Hubbard U and potentials: differs from row to row).
// The whole CSR (5.8e9 entries at BASELINE config 2) is held as T + C + one dictionary code per row for D: 0.17 GB.
// x += H y is two kernels, each walking the vector in the order that lets its gathers be served on chip:
//   k_pb_down  block couplings.  The rows y[b'][.] a block needs are ~17 OTHER blocks, each 103 KB -- read in block order they
//              come from HBM every time (measured in round 1: 23 of the 31 GB a product moved).  Here the vector is walked
//              PANEL-major: a panel = 16 consecutive positions (one 128-byte line; rows are pitched to a multiple of 16) of EVERY
//              block = N_dn lines = 1.6 MB, which stays in one XCD's 4 MiB L2 while its ~17 re-reads happen.
//   k_pb_up    in-block part + diagonal.  One workgroup stages a block's row of y in LDS (103 KB) and gathers from there; the
//              template T is stored per 64-row slice as 16-bit LDS indices, grouped by value (no value decode in the loop) and
//              edge-coloured on the host so that the 32 lanes of a half-wave hit 32 different LDS banks in every slot.
// Vectors are PITCHED: block b starts at element b*pitch, pitch = N_up rounded up to a multiple of 16 (padding stays zero).
#pragma once
#include "lpp_kernels.h"

namespace lpp {

constexpr int kPbMaxGroups = 8; // distinct off-diagonal values of the in-block matrix
constexpr int kPbZeroSlots = 32; // zero-valued window elements behind the row (one per LDS bank) that padding entries read
constexpr int kPbUpThreads = 1024;

// ---------------------------------------------------------------------------------------------
// in-block part + diagonal:  x[b][i] = beta' x[b][i] + alpha ( sum_k T[i][c_k] y[b][c_k] + D[b][i] y[b][i] )   (+ Re<y|x> partial)
// ---------------------------------------------------------------------------------------------
struct PbUpArgs {
	// template: for slice j and value group g, tw_len[j*G+g] chunks starting at chunk tw_off[j*G+g]; a chunk is 64 lanes x 2
	// words = 4 slots, each a 16-bit window index (low half first); filling entries index a zero slot
	const uint32_t* tw;
	const int32_t* tw_off;
	const uint16_t* tw_len;
	int G;
	double gval[kPbMaxGroups];
	const double* dict; // 256 doubles (diagonal codes)
	const uint8_t* dcode; // one code per row, pitched like the vectors
	int64_t n_up, pitch, n_blk;
	int spb; // slices per block
	const double* y;
	double* u; // out: alpha (T y + D y), pitched
	double* partial; // per-workgroup Re<y|u> (null: not wanted)
	EpiScale sc; // only alpha is used (chained form: alpha and beta)
	// chained form (KC > 0), see k_pb_up: wbuf holds w_{j-1} and receives r_j, ybuf holds r_{j-1} and receives the in-block part of w_j
	double* wbuf;
	double* ybuf;
	const double* g_a; // raw_{j-1} (null: the vector in wbuf is r_j already, nothing to subtract)
	const double* g_b2; // b_{j-2}^2
};

constexpr int kPbPre = 4; // chunks (of 4 slots) of every (slice, group) requested one slice ahead

// A wave's vector-memory results return IN ORDER (s_waitcnt vmcnt): a wait for an L2 load is also a wait for every HBM load
// issued before it, whatever the look-ahead (measured on a version that read-modify-wrote x here: 77 % of the wave cycles
// spent waiting; 47 % with a four-slice look-ahead, which the in-order queue defeats).  So this kernel touches HBM only in
// bulk: the block's row of y and its diagonal codes are staged in LDS in front of the barrier that waits for them anyway,
// the result u is stored and never read here, and x is not touched at all (the streaming pass that follows a product forms
// x = beta x + u + z).  Inside the slice loop only template words are loaded (L2 hits), one slice ahead, into two fixed
// register sets (a copy would wait for the load).  An entry is a 16-bit window index: one SDWA shift (index * 8 = LDS byte
// address, the window sits at LDS address 0), one ds_read_b64, one v_add_f64.  Both ends of that were measured: with a
// plain and/shift/add unpacking the kernel was VALU-bound (10 vector instructions per pair of entries); with ready-made
// 32-bit addresses the template words (19 GB per product through a ~57 GB/s-per-CU L2->L1 path) bound it instead.
// One-window form: the chunks of a list are stored in PAIRS (lane l's words of chunks 2q and 2q+1 side by side, pb_build) and requested with
// one 16-byte load per pair.  A wave-level load of 8 or of 16 bytes per lane occupies a CU's vector-memory pipeline for the same ~17 cycles
// (scripts/experiments/r04_tcp_rate.hip: 70 against 130-140 GB/s per CU), and the number of template loads was what the gather phase of
// this kernel waited for.  Look-ahead depths are then even.
#ifndef LPP_PB_PAIRS
#define LPP_PB_PAIRS 1
#endif
constexpr bool kPbPairs = LPP_PB_PAIRS != 0;
constexpr int kPbPreMax = 6; // deepest look-ahead of one value group (k_pb_up's PRE0)
// look-ahead depth of value group g (two groups share 8 chunks as PRE0 + (8 - PRE0); with pairs PRE0 = 6 leaves group 1 four)
__host__ __device__ constexpr int pb_depth(int GG, int g, int PRE0) { return GG == 2 ? (g == 0 ? PRE0 : (kPbPairs && PRE0 >= 6 ? 4 : 2 * kPbPre - PRE0)) : kPbPre; }
constexpr int kPbPreMin = 1; // chunks of a group requested whatever its list length (no branch)
template <int GT> struct PbWords {
	uint2 w[GT][kPbPreMax]; // only the first `depth(g)` chunks of group g are ever touched (the others take no register)
	int nc[GT];
	double yo; // chained form: the previous Lanczos vector at this lane's row (its beta term rides in u)
};

// index * 8 of the low / high 16 bits of w in ONE instruction (the compiler emits and + shift)
__device__ __forceinline__ uint32_t pb_lo8(uint32_t w)
{
	uint32_t r;
	asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(3u), "v"(w));
	return r;
}
__device__ __forceinline__ uint32_t pb_hi8(uint32_t w)
{
	uint32_t r;
	asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(3u), "v"(w));
	return r;
}
// LDS read at an absolute byte address (k_pb_up owns the whole LDS allocation: its dynamic array starts at address 0)
typedef __attribute__((address_space(3))) const double pb_lds_cdouble;
__device__ __forceinline__ double pb_lds_abs(uint32_t byte_addr) { return *(pb_lds_cdouble*)(uintptr_t)byte_addr; }

// LDS layout (all dynamic, so that the window starts at LDS address 0 and a template word IS the address):
//   [0, (pitch+32)*8) window + zero slots | dcode_s[pitch] | off_s[spb*G] | len_s[spb*G] | dict_s[256] | smem[16] | (pad)
__host__ __device__ inline size_t pb_up_meta_offset(int64_t pitch) { return sizeof(double) * (size_t)(pitch + kPbZeroSlots) + (size_t)pitch; }
__host__ __device__ inline size_t pb_up_dict_offset(int64_t pitch, int spb, int G) { return (pb_up_meta_offset(pitch) + (size_t)spb * (size_t)G * 6 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pb_up_lds_bytes(int64_t pitch, int spb, int G)
{
	return pb_up_dict_offset(pitch, spb, G) + 256 * sizeof(double) + (kPbUpThreads / 64) * sizeof(double) + 16;
}

// GT = number of value groups (1 or 2: unrolled, with look-ahead; 0: any G <= 8, plain loop).
// CHAIN: the chained form of the scale-free Lanczos step (pb_launch_chain).  The previous step left w = H r/b - (b/b') r' complete
// in wbuf and its own vector r in ybuf, but did NOT run the pass  r_next = w - g r  (g = raw / b^2): this kernel does it while it
// stages the row -- it reads both rows, keeps r_next in the LDS window and writes it back over w (k_pb_down gathers from there).
// Its result u = alpha (T r_next + D r_next) goes to the buffer u as always; k_pb_down<RMW> then forms the new
//   w = u + alpha C r_next   over r in ybuf  (u = alpha (T + D) r_next + beta r).
// A step is 8 passes over the vector (here: 3 reads -- w, r and r again for the beta term, which rides in u --, 2 writes; there: the
// gathers, 1 read, 1 write) in two launches, instead of 9 in three with the separate combine pass (2 + 2 + 5).  (Keeping the old row in
// registers for the beta term here would save the re-read, but 2 x 13 registers on top of the look-ahead words spill: 476 bytes of
// scratch per lane at KC = 14.)
// PRE0 (two value groups): chunks of group 0 requested one slice ahead; group 1 gets 2 kPbPre - PRE0.  The lists of the two
// groups are not equally long (config 2: hops with sign + average 11.7 entries per row, 4-5 chunks; sign -: 5.4 entries, 2-3
// chunks), and a chunk beyond the look-ahead is a load with a full L2 round trip in the middle of a slice.  The host picks
// PRE0 from the template's list lengths (pb_build).
template <bool DOT, int GT, bool CHAIN = false, int PRE0 = kPbPre> __global__ __launch_bounds__(kPbUpThreads, 4) void k_pb_up(PbUpArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // pitch + kPbZeroSlots elements, at LDS address 0
	uint8_t* dcode_s = (uint8_t*)(win + a.pitch + kPbZeroSlots); // [pitch]
	int32_t* off_s = (int32_t*)(lds_raw + pb_up_meta_offset(a.pitch)); // [spb*G]
	uint16_t* len_s = (uint16_t*)(off_s + a.spb * a.G); // [spb*G]
	double* dict_s = (double*)(lds_raw + pb_up_dict_offset(a.pitch, a.spb, a.G));
	double* smem = dict_s + 256;
	for (int i = threadIdx.x; i < 256; i += kPbUpThreads) dict_s[i] = a.dict[i];
	for (int i = threadIdx.x; i < a.spb * a.G; i += kPbUpThreads) {
		off_s[i] = a.tw_off[i];
		len_s[i] = a.tw_len[i];
	}
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	double gco = 0.0; // chained form: r_next = w - gco r
	if (CHAIN && a.g_a) {
		gco = *a.g_a;
		const double b2 = *a.g_b2;
		if (sqrt(b2) >= 1e-10) gco /= b2;
	}
	constexpr int NW = kPbUpThreads / 64;
	constexpr int GG = GT > 0 ? GT : 1;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int p2 = (int)(a.pitch >> 1), p16 = (int)(a.pitch >> 4); // pitch is a multiple of 16
	double dot = 0.0;
	// plain locals for everything the stage functions touch: a reference to the kernel-argument struct inside a lambda made
	// hipcc keep a private copy of it in scratch memory
	const uint2* const tw2 = (const uint2*)a.tw;
	const uint4* const tw4 = (const uint4*)a.tw;
	double* const uout = a.u;
	const int spb = a.spb, n_up = (int)a.n_up, G = a.G;
	double gv[GG];
#pragma unroll
	for (int g = 0; g < GG; g++) gv[g] = a.gval[g];
	auto gather4 = [=](const uint2& w, double& s0, double& s1) __attribute__((always_inline)) {
		s0 += pb_lds_abs(pb_lo8(w.x));
		s1 += pb_lds_abs(pb_hi8(w.x));
		s0 += pb_lds_abs(pb_lo8(w.y));
		s1 += pb_lds_abs(pb_hi8(w.y));
	};
	for (int64_t blk = blockIdx.x; blk < a.n_blk; blk += gridDim.x) {
		const double2* yb = (const double2*)(a.y + blk * a.pitch);
		const uint4* db = (const uint4*)(a.dcode + blk * a.pitch);
		const int64_t rowbase = blk * a.pitch;
		__syncthreads(); // everyone is done with the previous window (and the metadata is in place)
		// stage the row of y (8 independent 16-byte loads per thread in flight, branch-free: indices beyond the row are clamped,
		// those threads re-load and re-store the last element) and the block's diagonal codes
		if (CHAIN) {
			double2* const wb = (double2*)(a.wbuf + rowbase);
			const double2* const yo = (const double2*)(a.ybuf + rowbase);
			constexpr int NS = 4; // pairs per thread and row in flight: 2 -> 1.58 ms, 4 -> 1.57 ms, 8 spills (3.5 ms)
			for (int i0 = threadIdx.x; i0 < p2; i0 += NS * kPbUpThreads) {
				// loads unconditional and clamped (all of them in flight together); stores only from the lane that owns the pair -- a
				// clamped lane would subtract g r a second time from a pair its owner has already updated in place
				double2 wv[NS], yv[NS];
				int idx[NS];
#pragma unroll
				for (int q = 0; q < NS; q++) idx[q] = min(i0 + q * kPbUpThreads, p2 - 1);
#pragma unroll
				for (int q = 0; q < NS; q++) wv[q] = wb[idx[q]];
				if (gco != 0.0) {
#pragma unroll
					for (int q = 0; q < NS; q++) yv[q] = yo[idx[q]];
#pragma unroll
					for (int q = 0; q < NS; q++) {
						wv[q].x -= gco * yv[q].x; // padding: 0 - g 0
						wv[q].y -= gco * yv[q].y;
						if (i0 + q * kPbUpThreads < p2) wb[idx[q]] = wv[q];
					}
				}
#pragma unroll
				for (int q = 0; q < NS; q++)
					if (i0 + q * kPbUpThreads < p2) ((double2*)win)[idx[q]] = wv[q];
			}
			for (int i0 = threadIdx.x; i0 < p16; i0 += kPbUpThreads) ((uint4*)dcode_s)[i0] = db[i0];
		} else {
			constexpr int NS = 8; // loads per thread and pass
			for (int pass = 0; pass < 8 / NS; pass++) {
				double2 t[NS];
				int idx[NS];
#pragma unroll
				for (int q = 0; q < NS; q++) idx[q] = min((int)threadIdx.x + (pass * NS + q) * kPbUpThreads, p2 - 1);
#pragma unroll
				for (int q = 0; q < NS; q++) t[q] = yb[idx[q]];
#pragma unroll
				for (int q = 0; q < NS; q++) ((double2*)win)[idx[q]] = t[q];
			}
			const int di = min((int)threadIdx.x, p16 - 1);
			((uint4*)dcode_s)[di] = db[di];
			for (int i0 = threadIdx.x + 8 * kPbUpThreads; i0 < p2; i0 += kPbUpThreads) ((double2*)win)[i0] = yb[i0]; // rows beyond 16384 elements
			for (int i0 = threadIdx.x + kPbUpThreads; i0 < p16; i0 += kPbUpThreads) ((uint4*)dcode_s)[i0] = db[i0];
		}
		if (threadIdx.x < kPbZeroSlots) win[a.pitch + threadIdx.x] = 0.0;
		__syncthreads();
		const double* const yold = CHAIN ? a.ybuf + rowbase : nullptr;
		// the row's own element and its diagonal value (code -> dictionary: two dependent LDS reads): asked for in FRONT of a slice's
		// gathers by the look-ahead path, so that their latency passes under the gathers instead of behind them
		auto own = [=](int j, double& yc, double& dv) __attribute__((always_inline)) {
			const int iu = min(j * 64 + lane, n_up - 1);
			yc = win[iu];
			dv = dict_s[dcode_s[iu]];
		};
		auto finish = [=, &dot](int j, double acc, double yc, double dv, double yo) __attribute__((always_inline)) {
			const int iu = j * 64 + lane;
			acc = fma(dv, yc, acc);
			if (iu < n_up) {
				// chained form: u = alpha (T + D) r_j + beta r_{j-1}, so that the coupling kernel needs neither r_{j-1} nor beta
				const double uv = CHAIN ? fma(beta, yo, alpha * acc) : alpha * acc;
				__builtin_nontemporal_store(uv, &uout[rowbase + iu]);
				if (DOT) dot += yc * uv;
			}
		};
		auto epilogue = [=](int j, double acc, double yo = 0.0) __attribute__((always_inline)) {
			double yc, dv;
			own(j, yc, dv);
			finish(j, acc, yc, dv, yo);
		};
		if (GT > 0) {
			auto load_words = [=](int j, PbWords<GG>& s) __attribute__((always_inline)) {
				if (j >= spb) return; // wave-uniform
#pragma unroll
				for (int g = 0; g < GG; g++) {
					s.nc[g] = __builtin_amdgcn_readfirstlane((int)len_s[j * GG + g]); // uniform: scalar branches below
					constexpr int PRE0_ = PRE0;
					const int depth = pb_depth(GG, g, PRE0_);
					if (kPbPairs) {
						const uint4* wp4 = tw4 + (size_t)off_s[j * GG + g] * 64 + lane; // offsets count pairs
#pragma unroll
						for (int c = 0; c < kPbPreMax; c += 2)
							if (c < depth && (c < kPbPreMin || c < s.nc[g])) {
								const uint4 t = wp4[(c >> 1) * 64];
								s.w[g][c] = uint2 { t.x, t.y };
								s.w[g][c + 1] = uint2 { t.z, t.w };
							}
					} else {
						const uint2* wp = tw2 + (size_t)off_s[j * GG + g] * 64 + lane;
#pragma unroll
						for (int c = 0; c < kPbPreMax; c++)
							if (c < depth && (c < kPbPreMin || c < s.nc[g])) s.w[g][c] = wp[c * 64]; // beyond the first chunks only what the list holds (scalar branch)
					}
				}
				// chained form: r_{j-1} at this row, re-read one slice ahead with the words.  By the counters the re-read does NOT hit L2
				// (the XCD streams ~10 MB between staging and here): 1.33 GB more fabric reads per step at config 2, served one slice
				// before it is used; k_pb_up pays 0.05 ms for it, k_pb_down<RMW> saves 0.14 ms by not reading r_{j-1} at all
				if (CHAIN) s.yo = yold[min(j * 64 + lane, n_up - 1)];
			};
			// one value group of a slice: sum of the window elements its (look-ahead) chunks index, longer lists streamed
			auto group_sum = [=](int j, int g, int nc, const uint2* w, int depth) __attribute__((always_inline)) {
				double s0 = 0.0, s1 = 0.0;
				constexpr int BT = 2; // chunks per batch: 4 BT LDS gathers in flight (3: no faster, scripts/experiments/README.md)
#pragma unroll
				for (int c = 0; c < kPbPreMax; c += BT) {
					if (c >= depth) continue;
					// the branch taken is the largest k <= BT with nc >= c + k (and c + k <= depth)
					if (BT >= 3 && c + 3 <= depth && nc >= c + 3) {
						gather4(w[c], s0, s1);
						gather4(w[c + 1], s0, s1);
						gather4(w[c + 2], s0, s1);
					} else if (c + 2 <= depth && nc >= c + 2) {
						gather4(w[c], s0, s1);
						gather4(w[c + 1], s0, s1);
					} else if (nc >= c + 1) {
						gather4(w[c], s0, s1);
					}
				}
				if (nc > depth) {
					if (kPbPairs) {
						const uint4* wp4 = tw4 + (size_t)off_s[j * GG + g] * 64 + lane;
						for (int c = depth; c < nc; c += 2) { // depth is even
							const uint4 t = wp4[(c >> 1) * 64];
							gather4(uint2 { t.x, t.y }, s0, s1);
							if (c + 1 < nc) gather4(uint2 { t.z, t.w }, s0, s1);
						}
					} else {
						const uint2* wp = tw2 + (size_t)off_s[j * GG + g] * 64 + lane;
						for (int c = depth; c < nc; c++) {
							const uint2 wr = wp[c * 64];
							gather4(wr, s0, s1);
						}
					}
				}
				return s0 + s1;
			};
			auto compute = [=](int j, const PbWords<GG>& s) __attribute__((always_inline)) {
				if (j >= spb) return; // wave-uniform
				double yc, dv;
				own(j, yc, dv);
				double acc = 0.0;
#pragma unroll
				for (int g = 0; g < GG; g++) acc = fma(gv[g], group_sum(j, g, s.nc[g], s.w[g], pb_depth(GG, g, PRE0)), acc);
				finish(j, acc, yc, dv, CHAIN ? s.yo : 0.0);
			};
			PbWords<GG> wa, wb;
			load_words(wave, wa);
			for (int j0 = wave; j0 < spb; j0 += 2 * NW) {
				load_words(j0 + NW, wb);
				compute(j0, wa);
				load_words(j0 + 2 * NW, wa);
				compute(j0 + NW, wb);
			}
		} else {
			for (int j = wave; j < spb; j += NW) {
				double acc = 0.0;
				// any number of value groups (<= 8; complex hoppings realified: 4): the words of a list are requested four chunks at a time
				// and the first four of the NEXT group before this group's gathers (one load per chunk, each waited for, was a full L2 round
				// trip per chunk: 2.6 ms at 6.4e7 complex states against 1.1 ms for the coupling kernel)
				const double yo_g = CHAIN ? yold[min(j * 64 + lane, n_up - 1)] : 0.0; // chained form: beta r_{j-1} rides in u (as in the unrolled paths)
				uint2 wn[4];
				int ncn = __builtin_amdgcn_readfirstlane((int)len_s[j * G]);
				// four chunks of a list from chunk c0 on: four 8-byte loads, or two 16-byte loads of the pair layout
				auto load4 = [=](size_t off, int c0, int nc, uint2* w) __attribute__((always_inline)) {
					if (kPbPairs) {
						const uint4* wp4 = tw4 + off * 64 + lane;
#pragma unroll
						for (int q = 0; q < 4; q += 2)
							if (c0 + q < nc) {
								const uint4 t = wp4[((c0 + q) >> 1) * 64];
								w[q] = uint2 { t.x, t.y };
								w[q + 1] = uint2 { t.z, t.w };
							}
					} else {
						const uint2* wp = tw2 + off * 64 + lane;
#pragma unroll
						for (int q = 0; q < 4; q++)
							if (c0 + q < nc) w[q] = wp[(c0 + q) * 64];
					}
				};
				size_t offn = (size_t)off_s[j * G];
				load4(offn, 0, ncn, wn);
				for (int g = 0; g < G; g++) { // wave-uniform trip counts
					const int nc = ncn;
					const size_t off = offn;
					uint2 w[4];
#pragma unroll
					for (int q = 0; q < 4; q++) w[q] = wn[q];
					if (g + 1 < G) {
						ncn = __builtin_amdgcn_readfirstlane((int)len_s[j * G + g + 1]);
						offn = (size_t)off_s[j * G + g + 1];
						load4(offn, 0, ncn, wn);
					}
					double s0 = 0.0, s1 = 0.0;
#pragma unroll
					for (int q = 0; q < 4; q++)
						if (q < nc) gather4(w[q], s0, s1);
					for (int c0 = 4; c0 < nc; c0 += 4) { // longer lists: four more chunks in flight at a time
						uint2 wr[4];
						load4(off, c0, nc, wr);
#pragma unroll
						for (int q = 0; q < 4; q++)
							if (c0 + q < nc) gather4(wr[q], s0, s1);
					}
					acc = fma(a.gval[g], s0 + s1, acc);
				}
				epilogue(j, acc, yo_g);
			}
		}
	}
	if (DOT) {
		const double r = block_sum_n<kPbUpThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// block couplings:  z[b][i] = alpha sum_k C[b][b'_k] y[b'_k][i]   (+ Re<y|z> partial); the caller adds z to the x of k_pb_up
// One persistent workgroup per CU.  Workgroup w belongs to group w mod 8 (one XCD under round-robin dispatch: speed only) and
// owns a fixed range of blocks for ALL panels of its group; the couplings of its blocks sit in LDS (byte offsets of the source
// blocks + value codes), so nothing but y and z moves through L2.  Group k walks panels k, k+8, ...; the workgroups of a group
// stay within two panels of each other (bounded pacing: per-group, per-panel counters), which keeps the panel in that L2.
// A wave task covers 8 blocks x 16 positions with 16-byte lanes.  The blocks of a workgroup are taken in the order of
// decreasing list length (`order`), so the 8 blocks of a task have lists of (nearly) the same length and the task's trip count
// is its longest list rounded up to 4 -- padded places cost real L1 traffic.
// The kernel only READS y (L2 hits after the first touch of a line) and WRITES z (non-temporal): no HBM load sits in the
// in-order return queue in front of the gathers (an x read-modify-write here cost an HBM round trip per task).
// ---------------------------------------------------------------------------------------------
// LDS image of k_pb_down: one 32-bit word per place of a block's coupling list (source block | value code << 24), rows of
// `stride` words.  A lane reads the 4 places of a chunk with ONE 16-byte LDS read (the 8 blocks of a task sit `stride` words
// apart: stride / 4 odd puts their 16-byte segments on different bank groups); separate 2-byte index and 1-byte code arrays
// cost 8 LDS instructions per chunk instead, and the LDS pipeline was what the kernel's instruction skeleton waited for
// (0.8 of its 0.875 ms at config 2: scripts/experiments/r03_down_parts_ab.sh).
__host__ __device__ inline int pb_down_stride(int rowcap) { return ((rowcap >> 2) & 1) ? rowcap : rowcap + 4; }
__host__ __device__ inline size_t pb_down_lds_bytes(int ids_per_wg, int rowcap) { return (((size_t)ids_per_wg * 8 + 15) & ~(size_t)15) + (size_t)ids_per_wg * (size_t)pb_down_stride(rowcap) * 4 + 16; }

// The image of `nown` blocks order[b0 ...]: row[il] = block * rowmul, len[il] = its list length, place[il * stride + k] = source block | code << 24.
// Places beyond a list carry code 0 (+0.0) and the address the task's first block (the longest list) reads at this place, so the filling
// lanes of a gather ask for a line that is requested anyway; no per-lane conditions in the gather loop.
__device__ inline void pb_down_fill_image(uint32_t* row, int32_t* len, uint32_t* place, int stride, int rowcap, const int32_t* order, const int64_t* c_ptr,
                                          const int32_t* c_col, const uint8_t* c_code, int64_t b0, int nown, uint32_t rowmul, int tid, int nthreads)
{
	constexpr int BPT = 8; // blocks of a wave task
	for (int il = tid; il < nown; il += nthreads) {
		const int64_t b = order[b0 + il];
		row[il] = (uint32_t)b * rowmul;
		len[il] = (int32_t)(c_ptr[b + 1] - c_ptr[b]);
	}
	for (int i = tid; i < nown * rowcap; i += nthreads) {
		const int il = i / rowcap, k = i - il * rowcap;
		const int64_t b = order[b0 + il];
		const int64_t p0 = c_ptr[b];
		const bool in = k < (int)(c_ptr[b + 1] - p0);
		const int64_t bl = order[b0 + (il & ~(BPT - 1))];
		const int64_t pl = c_ptr[bl];
		const int32_t fill = k < (int)(c_ptr[bl + 1] - pl) ? c_col[pl + k] : (int32_t)bl;
		place[il * stride + k] = ((uint32_t)(in ? c_col[p0 + k] : fill) & 0xffffffu) | ((uint32_t)(in ? c_code[p0 + k] : (uint8_t)0) << 24);
	}
}

struct PbDownArgs {
	int64_t pitch, n_blk;
	int npanels; // pitch / 16
	int ids_per_wg; // blocks owned by one workgroup
	// More blocks per workgroup than one LDS image of their lists holds (round 5: sectors of 65536 blocks and more -- 77520 at the 4x5
	// lattice's (7,7) sector, 2423 per workgroup): the workgroup walks its range in `rounds` pieces of `ids_per_round` blocks (a multiple
	// of 64, the sorting window of `order`) inside every panel and rebuilds the image for each -- the lists come from L2, 4 % of what the
	// piece gathers.  rounds == 1: the image is built once per launch (ids_per_round == ids_per_wg)
	int rounds, ids_per_round;
	const uint4* image; // rounds > 1: the images of all (workgroup, round) pairs, made once (k_pb_down_image; rows as block numbers); null: rebuilt from the lists
	int rowcap; // longest coupling list rounded up to a multiple of 4
	const int64_t* c_ptr; // couplings: CSR over blocks, off-diagonal, ascending
	const int32_t* c_col;
	const uint8_t* c_code; // dictionary code of each coupling
	const int32_t* order; // [n_blk] blocks of every workgroup's range sorted by decreasing list length (global block numbers)
	const double* dict;
	const double* y; // addressed with 32-bit byte offsets (< 4 GiB)
	double* z;
	const double* u_in; // RMW: the in-block part of the product (k_pb_up's u)
	const double* shift; // RMW: s of the second partial |w - s y|^2 (k_b2_from_w)
	double* partial; // per-workgroup Re<y|z> (null: not wanted); RMW: pairs (Re<y|z>, |z|^2) of the finished z
	EpiScale sc; // only alpha is used: z = alpha * C y
	int* pace; // [8][npanels] finished-workgroup counters (zeroed before the launch); null: free-running
	int u_has_beta; // RMW: u_in already holds beta * (old z) (k_pb_up<CHAIN> adds it): z is written only, one stream less
	const double2* cdict; // CPLX: 256 complex coupling values (the codes of c_code index it; `dict` then only serves the diagonal)
	int pf_lead; // PF: batches of 64 blocks whose u lines are touched for the NEXT panel
};

// RMW (chained Lanczos step, see k_pb_up): z holds the previous Lanczos vector r' and receives the finished
//   w = u_in + beta r' + alpha C y;   the partials are Re<y|w> and |w - s y|^2 (k_b2_from_w).
// The two HBM loads per task this needs are only USED behind the gather loop.  Where they are issued (in front of the gathers,
// behind the first or the second chunk) moved the kernel by less than 4 % (1.86 / 1.94 / 1.93 ms at BASELINE
// config 2): it is bound by what goes through L1 -- 17 gathered lines + 3 streamed ones per line written -- not by the order.
// WIDE: vectors beyond 4 GiB (BASELINE config 5's sectors).  The LDS image then holds 128-byte LINE numbers instead of byte
// offsets and every address is formed in 64 bits: (line of the source block's row + panel) << 7.
// CPLX: complex hoppings.  The vector is complex, one element per 16-byte lane (a line = 8 positions; pitch and npanels still count
// doubles), the coupling values are complex (cdict) and a gather is multiplied as a complex number; everything else -- lines, panels,
// pacing, the partial sums (Re<y|z> is the real dot product of the doubles) -- is the real kernel.
template <int THREADS, bool RMW = false, bool WIDE = false, bool CPLX = false, bool PF = false> __global__ __launch_bounds__(THREADS) void k_pb_down(PbDownArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	__shared__ double dict_s[CPLX ? 512 : 256]; // CPLX: (re, im) pairs
	// LDS image of this workgroup's coupling lists in `order` (block numbers and value codes, 3 bytes per place: a workgroup of
	// k_pb_up must fit next to this one)
	const int stride = pb_down_stride(a.rowcap);
	uint32_t* row_s = (uint32_t*)lds_raw; // [ids_per_round] byte offset of the block itself
	int32_t* len_s = (int32_t*)(row_s + a.ids_per_round); // [ids_per_round] list length
	uint32_t* place_s = (uint32_t*)(lds_raw + (((size_t)a.ids_per_round * 8 + 15) & ~(size_t)15)); // [ids_per_round][stride] source block (< 2^24) | code << 24
	__shared__ double smem_d[THREADS / 64];
	__shared__ int task_s[3]; // PF: next task of the panels k, k + 1, k + 2 (mod 3)
	if (PF && threadIdx.x < 3) task_s[threadIdx.x] = 0;
	for (int i = threadIdx.x; i < (CPLX ? 512 : 256); i += THREADS) dict_s[i] = CPLX ? ((const double*)a.cdict)[i] : a.dict[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	double dot = 0.0, nrm = 0.0;
	const double sh = RMW ? *a.shift : 0.0;
	constexpr int LPL = 8; // 16-byte lanes per panel line
	constexpr int BPT = 64 / LPL; // blocks of a wave task
	constexpr int LSH = 7; // log2 of the panel line's bytes
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPL, c = lane & (LPL - 1);
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1; // groups the panels are dealt over
	const int grp = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int slot = nx == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
	const int nslots = (int)(gridDim.x / nx);
	const uint32_t rowbytes = WIDE ? (uint32_t)((a.pitch * 8) >> LSH) : (uint32_t)(a.pitch * 8); // WIDE: panel lines per row
	const int64_t b0w = (int64_t)slot * a.ids_per_wg; // this workgroup's range of `order`
	const int nown_w = (int)max((int64_t)0, min((int64_t)a.ids_per_wg, a.n_blk - b0w));
	const int rounds = a.rounds;
	int64_t b0 = b0w;
	int nown = rounds == 1 ? nown_w : 0;
	auto build_image = [&](int r) __attribute__((always_inline)) {
		b0 = b0w + (int64_t)r * a.ids_per_round;
		nown = rounds == 1 ? nown_w : max(0, min(a.ids_per_round, nown_w - r * a.ids_per_round));
		if (rounds > 1 && a.image) { // a copy of the prepared image: 16-byte pieces, the rows' block numbers become offsets on the way
			const int n4 = (int)(pb_down_lds_bytes(a.ids_per_round, a.rowcap) >> 4);
			const uint4* const src = a.image + ((size_t)slot * rounds + r) * (size_t)n4;
			for (int i = threadIdx.x; i < n4; i += THREADS) {
				uint4 v = src[i];
				if (4 * i < a.ids_per_round) { // (ids_per_round is a multiple of 8: a piece holds rows only or none)
					v.x *= rowbytes;
					v.y *= rowbytes;
					v.z *= rowbytes;
					v.w *= rowbytes;
				}
				((uint4*)lds_raw)[i] = v;
			}
			return;
		}
		pb_down_fill_image(row_s, len_s, place_s, stride, a.rowcap, a.order, a.c_ptr, a.c_col, a.c_code, b0, nown, rowbytes, (int)threadIdx.x, THREADS);
	};
	if (rounds == 1) build_image(0);
	__syncthreads();
	const char* ysrc = (const char*)a.y;
	for (int p = grp; p < a.npanels; p += nx) {
		if (a.pace && p >= grp + 2 * nx) {
			// bounded wait: the panel before the previous one must be finished by every workgroup of the group
			if (threadIdx.x == 0) {
				const int* cnt = a.pace + (int64_t)grp * a.npanels + (p - 2 * nx);
				for (int spin = 0; spin < 8192 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nslots; spin++)
					__builtin_amdgcn_s_sleep(8);
			}
			__syncthreads();
		}
		const uint32_t colb = WIDE ? (uint32_t)(c * 16) : (uint32_t)((p << LSH) + c * 16); // byte offset of this lane's two positions inside a row (WIDE: inside the panel line)
		// byte offset of (row `r` as stored in the LDS image, this lane's two positions of panel p)
		auto at = [=](uint32_t r) __attribute__((always_inline)) -> size_t { return WIDE ? (((size_t)(r + (uint32_t)p) << LSH) + colb) : (size_t)(r + colb); };
		// PF: the tasks of a panel are handed out by an LDS counter (three counters in rotation: the one of panel k + 2 is zeroed during
		// panel k, between the barriers that end panels k - 1 and k), and the LAST wave first touches this workgroup's lines of u -- from
		// the third batch of 64 blocks on for this panel, the first two batches for the group's next panel (a batch = 8 tasks; the first
		// tasks of a panel start before anything asked for now has arrived).  The tasks' own u loads are then L2 hits.  A wave's results
		// return in order, so an HBM round trip in front of a task's gathers is waited for by all of them: 0.33 ms of this kernel's 1.67
		// at config 2 were that.  Here ONE wave waits once per panel, and takes fewer tasks for it.  (Touching the next panel's y lines
		// the same way: slower, 3.32 against 3.02 ms per step -- 1.6 MB more in the 4 MB L2 that holds the panel.)
		const int pk = (p - grp) / nx; // the group's k-th panel
		constexpr int NP = 8;
		uint32_t pft[PF ? NP : 1];
		if (PF && RMW) { // (the chained form: one round per panel, the image of the whole range is in place)
			if (wave == THREADS / 64 - 1 && nown > 0) { // (a workgroup without blocks -- more slots than ranges -- has no line to touch: row_s[-1] is not its word)
				const bool nextp = p + nx < a.npanels;
#pragma unroll
				for (int k = 0; k < NP; k++) {
					const uint32_t r = row_s[min(lane + 64 * k, nown - 1)];
					const uint32_t pp = (uint32_t)(k < a.pf_lead && nextp ? p + nx : p);
					pft[k] = *(const uint32_t*)((const char*)a.u_in + (WIDE ? (((size_t)(r + pp)) << LSH) : (size_t)(r + (pp << LSH))));
				}
			}
		}
		for (int r = 0; r < rounds; r++) {
		if (rounds > 1) { // (wave-uniform) this piece of the workgroup's range: its image replaces the previous piece's
			__syncthreads();
			build_image(r);
			__syncthreads();
		}
		const int ngroups = (nown + BPT - 1) / BPT;
		// PF: the tasks of a step (panel, round) are handed out by an LDS counter -- three counters in rotation: the one of step q + 2 is zeroed during
		// step q, i.e. behind the barrier that ended step q - 1, which used it last
		const int q = pk * rounds + r;
		if (PF && threadIdx.x == 0) task_s[(q + 2) % 3] = 0;
		auto next_task = [&](int prev) __attribute__((always_inline)) -> int {
			if (!PF) return prev < 0 ? wave : prev + THREADS / 64;
			int g = 0;
			if (lane == 0) g = atomicAdd(&task_s[q % 3], 1);
			return __builtin_amdgcn_readfirstlane(g);
		};
		for (int g = next_task(-1); g < ngroups; g = next_task(g)) {
			const int il = min(g * BPT + sub, nown - 1);
			const bool valid = g * BPT + sub < nown;
			// trip count of the task: the longest list is the first one (decreasing order), in chunks of 4
			const int n4 = (__builtin_amdgcn_readfirstlane(len_s[g * BPT]) + 3) >> 2;
			const uint4* const prow = (const uint4*)(place_s + il * stride); // a chunk = 4 places = one 16-byte read
			double2 acc = double2 { 0.0, 0.0 };
			// three chunks of 4 gathers in flight; the chunk loop is unrolled by three with one fixed buffer per stage (a buffer
			// rotated by register copies would wait for the loads it holds)
			double2 ga[4], gb[4], gc[4];
			uint4 pa, pb, pc; // the chunks' places (their codes are needed when the gathers have come back)
			auto issue = [&](int ch, double2* gbuf, uint4& pw) __attribute__((always_inline)) {
				pw = prow[ch];
				const uint32_t w4[4] = { pw.x, pw.y, pw.z, pw.w };
#pragma unroll
				for (int q = 0; q < 4; q++) {
					gbuf[q] = *(const double2*)(ysrc + at((w4[q] & 0xffffffu) * rowbytes));
				}
			};
			auto consume = [&](const double2* gbuf, const uint4& pw) __attribute__((always_inline)) {
				const uint32_t w4[4] = { pw.x, pw.y, pw.z, pw.w };
#pragma unroll
				for (int q = 0; q < 4; q++) {
					if (CPLX) {
						const double2 v = ((const double2*)dict_s)[w4[q] >> 24];
						acc.x = fma(v.x, gbuf[q].x, fma(-v.y, gbuf[q].y, acc.x));
						acc.y = fma(v.x, gbuf[q].y, fma(v.y, gbuf[q].x, acc.y));
					} else {
						const double v = dict_s[w4[q] >> 24];
						acc.x = fma(v, gbuf[q].x, acc.x);
						acc.y = fma(v, gbuf[q].y, acc.y);
					}
				}
			};
			double2* const zp = (double2*)((char*)a.z + at(row_s[il]));
			double2 uo = double2 { 0.0, 0.0 }, xo = double2 { 0.0, 0.0 };
			if (RMW) { // used only behind the gather loop: no wait here
				uo = nt_load2((const double2*)((const char*)a.u_in + at(row_s[il])));
				if (!a.u_has_beta) xo = nt_load2(zp); // wave-uniform
			}
			if (n4 > 0) issue(0, ga, pa);
			if (n4 > 1) issue(1, gb, pb);
			for (int ch = 0; ch < n4; ch += 3) { // wave-uniform conditions
				if (ch + 2 < n4) issue(ch + 2, gc, pc);
				consume(ga, pa);
				if (ch + 1 < n4) {
					if (ch + 3 < n4) issue(ch + 3, ga, pa);
					consume(gb, pb);
				}
				if (ch + 2 < n4) {
					if (ch + 4 < n4) issue(ch + 4, gb, pb);
					consume(gc, pc);
				}
			}
			const double2 yown = *(const double2*)(ysrc + at(row_s[il])); // the panel is in L2
			if (valid) {
				acc.x = fma(alpha, acc.x, fma(beta, xo.x, uo.x));
				acc.y = fma(alpha, acc.y, fma(beta, xo.y, uo.y));
				__builtin_nontemporal_store(acc.x, &zp->x);
				__builtin_nontemporal_store(acc.y, &zp->y);
				dot += yown.x * acc.x + yown.y * acc.y;
				if (RMW) {
					const double dx = acc.x - sh * yown.x, dy = acc.y - sh * yown.y;
					nrm += dx * dx + dy * dy;
				}
			}
		}
		}
		if (PF && RMW && wave == THREADS / 64 - 1 && nown > 0) { // (the touched lines are used by nobody here: this only keeps the loads)
			uint32_t x = 0;
#pragma unroll
			for (int k = 0; k < NP; k++) x ^= pft[k];
			asm volatile("" ::"v"(x));
		}
		if (a.pace || PF) {
			__syncthreads();
			if (a.pace && threadIdx.x == 0) __hip_atomic_fetch_add(a.pace + (int64_t)grp * a.npanels + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (a.partial) {
		const double r = block_sum_n<THREADS / 64>(dot, smem_d);
		if (threadIdx.x == 0) a.partial[RMW ? 2 * blockIdx.x : blockIdx.x] = r;
		if (RMW) {
			const double q = block_sum_n<THREADS / 64>(nrm, smem_d);
			if (threadIdx.x == 0) a.partial[2 * blockIdx.x + 1] = q;
		}
	}
}

// rounds > 1: the images of all (workgroup, round) pairs in the layout k_pb_down keeps in LDS, rows as block numbers.  grid = slots * rounds
static __global__ __launch_bounds__(256) void k_pb_down_image(PbDownArgs a, uint4* out)
{
	const int slot = (int)blockIdx.x / a.rounds, r = (int)blockIdx.x - slot * a.rounds;
	const int64_t b0w = (int64_t)slot * a.ids_per_wg;
	const int nown_w = (int)max((int64_t)0, min((int64_t)a.ids_per_wg, a.n_blk - b0w));
	const int nown = max(0, min(a.ids_per_round, nown_w - r * a.ids_per_round));
	const size_t bytes = pb_down_lds_bytes(a.ids_per_round, a.rowcap);
	unsigned char* const img = (unsigned char*)out + (size_t)blockIdx.x * bytes;
	uint32_t* const row = (uint32_t*)img;
	int32_t* const len = (int32_t*)(row + a.ids_per_round);
	uint32_t* const place = (uint32_t*)(img + (((size_t)a.ids_per_round * 8 + 15) & ~(size_t)15));
	pb_down_fill_image(row, len, place, pb_down_stride(a.rowcap), a.rowcap, a.order, a.c_ptr, a.c_col, a.c_code, b0w + (int64_t)r * a.ids_per_round, nown, 1u, (int)threadIdx.x, 256);
}

// leaving the chained form: the pending pass  y = y - g x  (g = *g_a / *g_b2), with the partials of Re<y_new|x> that the
// three-kernel form carries as <y | x_old>
static __global__ __launch_bounds__(kBlock) void k_pb_materialise(double2* __restrict__ y, const double2* __restrict__ x, const double* __restrict__ g_a,
                                                                  const double* __restrict__ g_b2, int64_t n2, double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	double g = *g_a;
	const double b2 = *g_b2;
	if (sqrt(b2) >= 1e-10) g /= b2;
	double c = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const double2 xv = x[i];
		double2 yv = y[i];
		yv.x -= g * xv.x;
		yv.y -= g * xv.y;
		y[i] = yv;
		c += yv.x * xv.x + yv.y * xv.y;
	}
	const double r = block_sum(c, smem);
	if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// The streaming pass behind a product.  The two product kernels leave u = alpha (T y + D y) and z = alpha C y; the new x is
//   x = beta x + u + z - g y        g = 0 (plain product: x = beta x + alpha H y), or the scale-free Lanczos coefficient
//                                   a_j / b_{j-1}^2 read from device memory (then this is k_axpy_nrm folded into the same pass)
// with partial sums of |x|^2 and of Re<x|y> (the latter is the next step's <y | beta x_old>, which the product kernels cannot
// form because they never read x).
struct PbCombineArgs {
	double2* x;
	const double2 *y, *u, *z;
	// the diagonal as a plain f64 stream (more than 256 distinct values: site-dependent U / potentials): the in-block kernel then
	// adds nothing for it (all codes 0 = +0.0) and this pass adds alpha D y; partial_dq receives the partials of sum D x_new^2,
	// the <y | D y> of the NEXT step's a_j (the product kernels' partials no longer hold it).  null: the diagonal travels as codes
	const double2* d;
	double* partial_dq;
	int64_t n2;
	EpiScale sc; // beta
	const double* a_ptr; // null: g = 0
	const double* b2_prev; // g = *a_ptr / *b2_prev (unless tiny), as in k_axpy_nrm
	double* partial_nrm; // null: no reductions
	double* partial_xy;
};

static __global__ __launch_bounds__(kBlock) void k_pb_combine(PbCombineArgs a)
{
	__shared__ double smem[kBlock / 64];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	const bool hasd = a.d != nullptr;
	const bool hasz = a.z != nullptr; // null: a matrix without block couplings (one block: pb_chain)
	double g = 0.0;
	if (a.a_ptr) {
		g = *a.a_ptr;
		if (a.b2_prev) {
			const double b2 = *a.b2_prev;
			if (sqrt(b2) >= 1e-10) g /= b2;
		}
	}
	double s = 0.0, c = 0.0, q = 0.0;
	const int64_t stride = (int64_t)gridDim.x * kBlock;
	int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
	constexpr int U = 4; // elements per lane and stream in flight (16 x 16-byte loads); measured with scripts/experiments/calib_combine.hip
	for (; i + (U - 1) * stride < a.n2; i += U * stride) {
		double2 xv[U], yv[U], uv[U], zv[U];
#pragma unroll
		for (int k = 0; k < U; k++) xv[k] = nt_load2(&a.x[i + k * stride]);
#pragma unroll
		for (int k = 0; k < U; k++) yv[k] = nt_load2(&a.y[i + k * stride]);
#pragma unroll
		for (int k = 0; k < U; k++) uv[k] = nt_load2(&a.u[i + k * stride]);
#pragma unroll
		for (int k = 0; k < U; k++) zv[k] = hasz ? nt_load2(&a.z[i + k * stride]) : double2 { 0.0, 0.0 };
		double2 dv[U];
		if (hasd) {
#pragma unroll
			for (int k = 0; k < U; k++) dv[k] = nt_load2(&a.d[i + k * stride]);
		}
#pragma unroll
		for (int k = 0; k < U; k++) {
			double2 r;
			r.x = beta * xv[k].x + uv[k].x + zv[k].x - g * yv[k].x;
			r.y = beta * xv[k].y + uv[k].y + zv[k].y - g * yv[k].y;
			if (hasd) {
				r.x = fma(alpha * dv[k].x, yv[k].x, r.x);
				r.y = fma(alpha * dv[k].y, yv[k].y, r.y);
				q += dv[k].x * r.x * r.x + dv[k].y * r.y * r.y;
			}
			nt_store2(r, &a.x[i + k * stride]);
			s += r.x * r.x + r.y * r.y;
			c += r.x * yv[k].x + r.y * yv[k].y;
		}
	}
	for (; i < a.n2; i += stride) {
		const double2 xv = a.x[i], yv = a.y[i], uv = a.u[i], zv = hasz ? a.z[i] : double2 { 0.0, 0.0 };
		double2 r;
		r.x = beta * xv.x + uv.x + zv.x - g * yv.x;
		r.y = beta * xv.y + uv.y + zv.y - g * yv.y;
		if (hasd) {
			const double2 dv = a.d[i];
			r.x = fma(alpha * dv.x, yv.x, r.x);
			r.y = fma(alpha * dv.y, yv.y, r.y);
			q += dv.x * r.x * r.x + dv.y * r.y * r.y;
		}
		a.x[i] = r;
		s += r.x * r.x + r.y * r.y;
		c += r.x * yv.x + r.y * yv.y;
	}
	if (a.partial_nrm) {
		const double rs = block_sum(s, smem);
		if (threadIdx.x == 0) a.partial_nrm[blockIdx.x] = rs;
		const double rc = block_sum(c, smem);
		if (threadIdx.x == 0) a.partial_xy[blockIdx.x] = rc;
	}
	if (a.partial_dq) {
		const double rq = block_sum(q, smem);
		if (threadIdx.x == 0) a.partial_dq[blockIdx.x] = rq;
	}
}

// Several GPUs, transposition exchange: the streaming pass behind the second all-to-all.  recv holds, for every rank p, the block
// couplings' part of this rank's blocks at the up-index range of p: chunk p = [own block][position in p's range], rows of pitch_dn.
//   x = beta x + u + (that part, back in block-major order);   partials: pairs (Re<y|x>, |x - s y|^2)  (k_b2_from_w)
// All indices in units of 16 bytes (pitches are multiples of 16 elements, so a pair never straddles two ranks' ranges; positions
// beyond N_up hold zeros on both sides).
static __global__ __launch_bounds__(kBlock) void k_pb_unpack_combine(double2* __restrict__ x, const double2* __restrict__ y, const double2* __restrict__ u,
                                                                     const double2* __restrict__ recv, int64_t nblk, int64_t pitch2, int64_t peru2, int64_t chunk2,
                                                                     EpiScale sc, double* __restrict__ partial, const double* __restrict__ shift,
                                                                     const double2* __restrict__ dplain = nullptr)
{
	__shared__ double smem[kBlock / 64];
	double alpha, beta;
	epi_coeffs(sc, alpha, beta);
	const double sh = shift ? *shift : 0.0;
	const int64_t n2 = nblk * pitch2;
	double dot = 0.0, nrm = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const int64_t b = i / pitch2, c = i - b * pitch2;
		const int64_t p = c / peru2, cl = c - p * peru2;
		const double2 zv = recv[p * chunk2 + b * peru2 + cl];
		const double2 xv = x[i], uv = u[i], yv = y[i];
		double2 r;
		r.x = beta * xv.x + uv.x + zv.x;
		r.y = beta * xv.y + uv.y + zv.y;
		if (dplain) { // the diagonal as a plain stream (see PbCombineArgs::d)
			const double2 dv = dplain[i];
			r.x = fma(alpha * dv.x, yv.x, r.x);
			r.y = fma(alpha * dv.y, yv.y, r.y);
		}
		x[i] = r;
		dot += yv.x * r.x + yv.y * r.y;
		const double dx = r.x - sh * yv.x, dy = r.y - sh * yv.y;
		nrm += dx * dx + dy * dy;
	}
	const double rd = block_sum(dot, smem);
	if (threadIdx.x == 0) partial[2 * blockIdx.x] = rd;
	const double rn = block_sum(nrm, smem);
	if (threadIdx.x == 0) partial[2 * blockIdx.x + 1] = rn;
}

// <y | D y> of a plain-stream diagonal for the vector a run starts from (afterwards the combine pass carries it along)
static __global__ __launch_bounds__(kBlock) void k_pb_dq(const double2* __restrict__ y, const double2* __restrict__ d, int64_t n2, double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	double q = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const double2 yv = y[i], dv = d[i];
		q += dv.x * yv.x * yv.x + dv.y * yv.y * yv.y;
	}
	const double r = block_sum(q, smem);
	if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// a_j of the scale-free recurrence from the two product kernels' partials and the carried <y | x_old>:
//   out = sum_p partial[p] + beta * (*xy)        (single block, fixed summation order)
static __global__ __launch_bounds__(kBlock) void k_pb_reduce_a(const double* __restrict__ partial, int np, const double* __restrict__ xy, EpiScale sc,
                                                               double* __restrict__ out)
{
	__shared__ double smem[kBlock / 64];
	double alpha, beta;
	epi_coeffs(sc, alpha, beta);
	double s = 0.0;
	for (int p = threadIdx.x; p < np; p += kBlock) s += partial[p];
	const double r = block_sum(s, smem);
	if (threadIdx.x == 0) out[0] = r + beta * xy[0] + alpha * xy[1]; // xy[1]: <y | D y> of a plain-stream diagonal (0 otherwise)
}

// ---------------------------------------------------------------------------------------------
// one-off kernels of the layout
// ---------------------------------------------------------------------------------------------

// plain CSR order of the whole matrix from (T, C, diagonal codes): lpp_engine_get_csr.  One thread per row.
// A row (b, i) holds, by ascending column: couplings to blocks b' < b, in-block entries with column < i, the diagonal,
// in-block entries with column > i, couplings to blocks b' > b -- exactly SparseRow::finalize's order (HubbardHelper.h:99).
static __global__ void k_pb_rebuild(int64_t n_up, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ t_ptr, const int32_t* __restrict__ t_col,
                                    const double* __restrict__ t_val, const int64_t* __restrict__ c_ptr, const int32_t* __restrict__ c_col,
                                    const uint8_t* __restrict__ c_code, const int64_t* __restrict__ blockbase, const uint8_t* __restrict__ dcode,
                                    const double* __restrict__ dict, int64_t* __restrict__ rowptr_out, int32_t* __restrict__ col_out,
                                    double* __restrict__ val_out, const double* __restrict__ dplain = nullptr, const int32_t* __restrict__ inv = nullptr)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t n = n_up * n_blk;
	if (r == 0 && rowptr_out) rowptr_out[n] = blockbase[n_blk];
	if (r >= n) return;
	const int64_t b = r / n_up, i = r - b * n_up;
	const int64_t c0 = c_ptr[b], c1 = c_ptr[b + 1];
	int64_t o = blockbase[b] + t_ptr[i] + i * (1 + (c1 - c0));
	if (rowptr_out) rowptr_out[r] = o;
	if (!col_out) return;
	int64_t p = c0;
	for (; p < c1 && c_col[p] < b; p++, o++) {
		col_out[o] = (int32_t)((int64_t)c_col[p] * n_up + i);
		val_out[o] = dict[c_code[p]];
	}
	int64_t q = t_ptr[i];
	const int64_t q1 = t_ptr[i + 1];
	for (; q < q1 && t_col[q] < i; q++, o++) {
		col_out[o] = (int32_t)(b * n_up + t_col[q]);
		val_out[o] = t_val[q];
	}
	col_out[o] = (int32_t)r;
	const int64_t at = b * pitch + (inv ? inv[i] : i); // where the layout keeps (b, i)
	val_out[o] = dplain ? dplain[at] : dict[dcode[at]];
	o++;
	for (; q < q1; q++, o++) {
		col_out[o] = (int32_t)(b * n_up + t_col[q]);
		val_out[o] = t_val[q];
	}
	for (; p < c1; p++, o++) {
		col_out[o] = (int32_t)((int64_t)c_col[p] * n_up + i);
		val_out[o] = dict[c_code[p]];
	}
}


// the same walk for complex hoppings: T and the couplings carry complex values (t_val: (re, im) pairs; cdict), the layout keeps a block as
// 2 n_up real positions (pitch counts doubles), the diagonal is real
static __global__ void k_pb_rebuild_c(int64_t n_up, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ t_ptr, const int32_t* __restrict__ t_col,
                                    const double2* __restrict__ t_val, const int64_t* __restrict__ c_ptr, const int32_t* __restrict__ c_col,
                                    const uint8_t* __restrict__ c_code, const int64_t* __restrict__ blockbase, const uint8_t* __restrict__ dcode,
                                    const double* __restrict__ dict, const double2* __restrict__ cdict, int64_t* __restrict__ rowptr_out, int32_t* __restrict__ col_out,
                                    double2* __restrict__ val_out, const double* __restrict__ dplain = nullptr, const int32_t* __restrict__ inv = nullptr)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t n = n_up * n_blk;
	if (r == 0 && rowptr_out) rowptr_out[n] = blockbase[n_blk];
	if (r >= n) return;
	const int64_t b = r / n_up, i = r - b * n_up;
	const int64_t c0 = c_ptr[b], c1 = c_ptr[b + 1];
	int64_t o = blockbase[b] + t_ptr[i] + i * (1 + (c1 - c0));
	if (rowptr_out) rowptr_out[r] = o;
	if (!col_out) return;
	int64_t p = c0;
	for (; p < c1 && c_col[p] < b; p++, o++) {
		col_out[o] = (int32_t)((int64_t)c_col[p] * n_up + i);
		val_out[o] = cdict[c_code[p]];
	}
	int64_t q = t_ptr[i];
	const int64_t q1 = t_ptr[i + 1];
	for (; q < q1 && t_col[q] < i; q++, o++) {
		col_out[o] = (int32_t)(b * n_up + t_col[q]);
		val_out[o] = t_val[q];
	}
	col_out[o] = (int32_t)r;
	const int64_t at = b * pitch + 2 * i; // the real part's position (the imaginary part carries the same code)
	val_out[o] = double2 { dplain ? dplain[at] : dict[dcode[at]], 0.0 };
	o++;
	for (; q < q1; q++, o++) {
		col_out[o] = (int32_t)(b * n_up + t_col[q]);
		val_out[o] = t_val[q];
	}
	for (; p < c1; p++, o++) {
		col_out[o] = (int32_t)((int64_t)c_col[p] * n_up + i);
		val_out[o] = cdict[c_code[p]];
	}
}


// ---------------------------------------------------------------------------------------------
// (T, C, D) from a CSR that was handed over (lpp_engine_set_csr: DefaultSymmetry.h:54-57 -> InternalProductStored.h:116), and the
// proof that the CSR is nothing but them
// ---------------------------------------------------------------------------------------------

// per block b: entries of its FIRST row (b, 0) that leave the block -- the couplings of block b.  FILL == false: count only.
// bad is raised when such an entry does not land on position 0 of another block.
template <bool FILL, typename V = double>
static __global__ void k_pb_csr_couplings(int64_t n_up, int64_t n_blk, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                          const V* __restrict__ val, int64_t* __restrict__ c_len, const int64_t* __restrict__ c_ptr,
                                          int32_t* __restrict__ c_col, V* __restrict__ c_val, int* __restrict__ bad)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blk) return;
	const int64_t r = b * n_up;
	int64_t n = 0;
	for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
		const int64_t c = col[p];
		if (c >= r && c < r + n_up) continue;
		if (c % n_up != 0) *bad = 1;
		if (FILL) {
			c_col[c_ptr[b] + n] = (int32_t)(c / n_up);
			c_val[c_ptr[b] + n] = val[p];
		}
		n++;
	}
	if (!FILL) c_len[b] = n;
}

// the diagonal of every row as a pitched f64 array; bad is raised for a row that stores no diagonal entry
static __global__ void k_pb_csr_diagonal(int64_t n_up, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                         const double* __restrict__ val, double* __restrict__ dval, int* __restrict__ bad)
{
	const int64_t n = n_up * n_blk;
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
		bool found = false;
		for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++)
			if ((int64_t)col[p] == r) {
				dval[(r / n_up) * pitch + (r % n_up)] = val[p];
				found = true;
				break;
			}
		if (!found) *bad = 1;
	}
}

// complex matrix: the stored diagonal must be real; both doubles of the position get it (pitch counts doubles)
static __global__ void k_pb_csr_diagonal_c(int64_t n_c, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                           const double2* __restrict__ val, double* __restrict__ dval, int* __restrict__ bad)
{
	const int64_t n = n_c * n_blk;
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
		bool found = false;
		for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++)
			if ((int64_t)col[p] == r) {
				const double2 v = val[p];
				if (__double_as_longlong(v.y) != 0) *bad = 1; // -0.0 would not be reproduced either
				const int64_t at = (r / n_c) * pitch + 2 * (r % n_c);
				dval[at] = v.x;
				dval[at + 1] = v.x;
				found = true;
				break;
			}
		if (!found) *bad = 1;
	}
}

static __global__ void k_pb_codes_from_values(int64_t n, const double* __restrict__ dval, const double* __restrict__ dict, int ndict, uint8_t* __restrict__ dcode)
{
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) dcode[k] = (uint8_t)dict_code(dict, ndict, dval[k]);
}

// every row of the CSR against the row (T, C, D) stand for, entry by entry and bit by bit (the walk of k_pb_rebuild)
static __global__ void k_pb_csr_verify(int64_t n_up, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ t_ptr, const int32_t* __restrict__ t_col,
                                       const double* __restrict__ t_val, const int64_t* __restrict__ c_ptr, const int32_t* __restrict__ c_col,
                                       const uint8_t* __restrict__ c_code, const int64_t* __restrict__ blockbase, const uint8_t* __restrict__ dcode,
                                       const double* __restrict__ dict, const double* __restrict__ dplain, const int64_t* __restrict__ rowptr,
                                       const int32_t* __restrict__ col, const double* __restrict__ val, int* __restrict__ bad,
                                       const int32_t* __restrict__ inv = nullptr)
{
	const int64_t n = n_up * n_blk;
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = r / n_up, i = r - b * n_up;
		const int64_t dat = b * pitch + (inv ? inv[i] : i);
		const int64_t c0 = c_ptr[b], c1 = c_ptr[b + 1];
		int64_t o = blockbase[b] + t_ptr[i] + i * (1 + (c1 - c0));
		const int64_t oe = o + (t_ptr[i + 1] - t_ptr[i]) + 1 + (c1 - c0);
		bool ok = rowptr[r] == o && rowptr[r + 1] == oe;
		if (ok) {
			auto same = [&](int64_t at, int64_t c, double v) { return (int64_t)col[at] == c && __double_as_longlong(val[at]) == __double_as_longlong(v); };
			int64_t p = c0;
			for (; p < c1 && c_col[p] < b; p++, o++) ok = ok && same(o, (int64_t)c_col[p] * n_up + i, dict[c_code[p]]);
			int64_t q = t_ptr[i];
			const int64_t q1 = t_ptr[i + 1];
			for (; q < q1 && t_col[q] < i; q++, o++) ok = ok && same(o, b * n_up + t_col[q], t_val[q]);
			ok = ok && same(o, r, dplain ? dplain[dat] : dict[dcode[dat]]);
			o++;
			for (; q < q1; q++, o++) ok = ok && same(o, b * n_up + t_col[q], t_val[q]);
			for (; p < c1; p++, o++) ok = ok && same(o, (int64_t)c_col[p] * n_up + i, dict[c_code[p]]);
		}
		if (!ok) *bad = 1;
	}
}

// the same check for complex hoppings (the walk of k_pb_rebuild_c)
static __global__ void k_pb_csr_verify_c(int64_t n_up, int64_t n_blk, int64_t pitch, const int64_t* __restrict__ t_ptr, const int32_t* __restrict__ t_col,
                                       const double2* __restrict__ t_val, const int64_t* __restrict__ c_ptr, const int32_t* __restrict__ c_col,
                                       const uint8_t* __restrict__ c_code, const int64_t* __restrict__ blockbase, const uint8_t* __restrict__ dcode,
                                       const double* __restrict__ dict, const double2* __restrict__ cdict, const double* __restrict__ dplain, const int64_t* __restrict__ rowptr,
                                       const int32_t* __restrict__ col, const double2* __restrict__ val, int* __restrict__ bad)
{
	const int64_t n = n_up * n_blk;
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = r / n_up, i = r - b * n_up;
		const int64_t dat = b * pitch + 2 * i;
		const int64_t c0 = c_ptr[b], c1 = c_ptr[b + 1];
		int64_t o = blockbase[b] + t_ptr[i] + i * (1 + (c1 - c0));
		const int64_t oe = o + (t_ptr[i + 1] - t_ptr[i]) + 1 + (c1 - c0);
		bool ok = rowptr[r] == o && rowptr[r + 1] == oe;
		if (ok) {
			auto same = [&](int64_t at, int64_t c, double2 v) { return (int64_t)col[at] == c && __double_as_longlong(val[at].x) == __double_as_longlong(v.x) && __double_as_longlong(val[at].y) == __double_as_longlong(v.y); };
			int64_t p = c0;
			for (; p < c1 && c_col[p] < b; p++, o++) ok = ok && same(o, (int64_t)c_col[p] * n_up + i, cdict[c_code[p]]);
			int64_t q = t_ptr[i];
			const int64_t q1 = t_ptr[i + 1];
			for (; q < q1 && t_col[q] < i; q++, o++) ok = ok && same(o, b * n_up + t_col[q], t_val[q]);
			ok = ok && same(o, r, double2 { dplain ? dplain[dat] : dict[dcode[dat]], 0.0 });
			o++;
			for (; q < q1; q++, o++) ok = ok && same(o, b * n_up + t_col[q], t_val[q]);
			for (; p < c1; p++, o++) ok = ok && same(o, (int64_t)c_col[p] * n_up + i, cdict[c_code[p]]);
		}
		if (!ok) *bad = 1;
	}
}

// pitched vector between the natural order of positions and the stored one: dst[b][p] = src[b][perm[p]] (TO_STORED) or
// dst[b][perm[p]] = src[b][p]; padding positions are written as zero
template <bool TO_STORED>
static __global__ void k_pb_permute(double* __restrict__ dst, const double* __restrict__ src, const int32_t* __restrict__ perm, int64_t n_blk, int64_t rows,
                                    int64_t pitch)
{
	const int64_t n = n_blk * pitch;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = k / pitch, p = k - b * pitch;
		if (p >= rows) {
			dst[k] = 0.0;
			continue;
		}
		if (TO_STORED) dst[k] = src[b * pitch + perm[p]];
		else dst[b * pitch + perm[p]] = src[k];
	}
}

// start vector in the pitched layout: element (b, i) takes the value the unpitched stream gives index b*rows + i
static __global__ void k_fill_random_pitched(double* __restrict__ v, int64_t n_blk, int64_t rows, int64_t pitch, int64_t offset, uint64_t seed,
                                             const int32_t* __restrict__ perm = nullptr)
{
	const int64_t n = n_blk * pitch;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = k / pitch, i = k - b * pitch;
		double val = 0.0;
		if (i < rows) {
			const uint64_t r = splitmix64(seed * 0x2545F4914F6CDD1DULL + (uint64_t)(b * rows + (perm ? perm[i] : i) + offset));
			val = (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5;
		}
		v[k] = val;
	}
}

} // namespace lpp
