// lpp_pbws_kernels.h -- the chained in-block kernel with its row staging taken off the gather waves (k_pb_up_ws).
//
// k_pb_up<CHAIN> alternates two phases per block: stage the row (read w and r, write r_next: 3 x 103 KB of HBM traffic per
// block at config 2) and gather from the LDS window.  One workgroup fills a CU (the window is 103 KB of the 160), so nothing
// overlaps the two: 15 us of staging + 16 us of gathers per block.  A second window does not fit in LDS -- but a row fits in
// REGISTERS: 5 of the 16 waves ("loaders": 320 lanes x 21 double2 = 107 KB) hold block k+1's row while the other 11 waves
// gather block k from the window.  Vector-memory results return in order PER WAVE, so the loaders' HBM loads sit in nobody's
// way: the gather waves still see L2 hits only.  Between two blocks the loaders copy their registers into the window
// (ds_write, two barriers).  The diagonal codes of block k+1 go straight into the second of two code buffers in LDS.
//
// MEASURED (config 2, scripts/experiments/r03_ws_ab.sh): 1.63 ms with the beta term left to k_pb_down, 1.86-1.99 ms with it
// here, against 1.56 ms for k_pb_up<CHAIN>; 6, 7 or 8 loader waves are slower still.  The gather phase is not hidden
// latency that fewer waves could carry: template words (L2->L1), LDS reads and vector issue are each 40-60 % busy during it,
// and it stretches by 16/11 when 5 waves leave it -- more than the 9 us of staging the overlap takes off a block.  The
// re-read of r_{j-1} for the beta term also stops being an L2 hit (the loaders stream 10 MB per XCD between the read and
// the re-read).  Opt-in (LPP_PB_WS=1), kept with its parity test as the record of that experiment.
#pragma once
#include "lpp_pb_kernels.h"

namespace lpp {

#ifndef LPP_PBWS_LOADERS
#define LPP_PBWS_LOADERS 5
#endif
#ifndef LPP_PBWS_NQ
#define LPP_PBWS_NQ 21
#endif
#ifndef LPP_PBWS_CH
#define LPP_PBWS_CH 3
#endif
#ifndef LPP_PBWS_AUX
#define LPP_PBWS_AUX 0 // cache policy bits of the loaders' buffer loads and stores (2: nt)
#endif
constexpr int kPbWsAux = LPP_PBWS_AUX;
constexpr int kPbWsLoaders = LPP_PBWS_LOADERS; // waves that stream the next block's row
constexpr int kPbWsGather = kPbUpThreads / 64 - kPbWsLoaders; // waves that gather
constexpr int kPbWsNQ = LPP_PBWS_NQ; // 16-byte pairs per loader lane
constexpr int kPbWsCh = LPP_PBWS_CH; // pairs of r requested together
constexpr int64_t kPbWsMaxPitch = 2 * (int64_t)kPbWsNQ * 64 * kPbWsLoaders; // 13440 elements

// LDS: window + zero slots | dcode_s[2][pitch] | off_s[spb*G] | len_s[spb*G] | dict_s[256]
__host__ __device__ inline size_t pb_ws_meta_offset(int64_t pitch) { return sizeof(double) * (size_t)(pitch + kPbZeroSlots) + 2 * (size_t)pitch; }
__host__ __device__ inline size_t pb_ws_dict_offset(int64_t pitch, int spb, int G) { return (pb_ws_meta_offset(pitch) + (size_t)spb * (size_t)G * 6 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t pb_ws_lds_bytes(int64_t pitch, int spb, int G) { return pb_ws_dict_offset(pitch, spb, G) + 256 * sizeof(double) + 16; }

// loaders: r_next = w - g r of one block into registers (and back over w), its diagonal codes into the LDS buffer `dc`.
// Buffer loads and stores through one descriptor per row: the address is descriptor + ONE lane offset register + a scalar
// offset per pair, and the range check of the descriptor stands in for the clamps (a load beyond the row returns 0, a store
// beyond it is dropped).  With flat addresses the compiler kept 21 clamped 64-bit addresses alive beside the 84 row registers
// and spilled up to 286 of them.
typedef uint32_t pb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 pb_ws_d2(pb_u32x4 v)
{
	double2 r;
	__builtin_memcpy(&r, &v, 16);
	return r;
}
__device__ __forceinline__ pb_u32x4 pb_ws_u4(double2 v)
{
	pb_u32x4 r;
	__builtin_memcpy(&r, &v, 16);
	return r;
}
__device__ __forceinline__ void pb_ws_fetch(double2 (&rr)[kPbWsNQ], double* wrow, const double* yrow, const uint8_t* drow, uint8_t* dc, int ll, int pitch, double gco)
{
	constexpr int NLB = kPbWsLoaders * 64 * 16; // bytes one pass of the loader lanes covers
	const auto wr = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, pitch * 8, 0x00020000);
	const auto yr = __builtin_amdgcn_make_buffer_rsrc((void*)yrow, 0, pitch * 8, 0x00020000);
	const auto dr = __builtin_amdgcn_make_buffer_rsrc((void*)drow, 0, pitch, 0x00020000);
	const int vo = ll * 16;
#pragma unroll
	for (int q = 0; q < kPbWsNQ; q++) rr[q] = pb_ws_d2(__builtin_amdgcn_raw_buffer_load_b128(wr, vo, q * NLB, kPbWsAux)); // all in flight together
	if (gco != 0.0) {
		// r in chunks, two chunks in flight (each costs an HBM round trip: 7 chunks one after the other are 7 round trips per block)
		constexpr int CH = kPbWsCh, NC = kPbWsNQ / kPbWsCh;
		double2 yv[2][CH];
#pragma unroll
		for (int q = 0; q < CH; q++) yv[0][q] = pb_ws_d2(__builtin_amdgcn_raw_buffer_load_b128(yr, vo, q * NLB, kPbWsAux));
#pragma unroll
		for (int c = 0; c < NC; c++) {
			if (c + 1 < NC) {
#pragma unroll
				for (int q = 0; q < CH; q++) yv[(c + 1) & 1][q] = pb_ws_d2(__builtin_amdgcn_raw_buffer_load_b128(yr, vo, ((c + 1) * CH + q) * NLB, kPbWsAux));
			}
#pragma unroll
			for (int q = 0; q < CH; q++) {
				rr[c * CH + q].x -= gco * yv[c & 1][q].x;
				rr[c * CH + q].y -= gco * yv[c & 1][q].y;
				__builtin_amdgcn_raw_buffer_store_b128(pb_ws_u4(rr[c * CH + q]), wr, vo, (c * CH + q) * NLB, kPbWsAux);
			}
			__builtin_amdgcn_sched_barrier(0); // no further look-ahead: hoisting all 21 loads costs 84 more registers
		}
	}
	for (int o = vo; o < pitch; o += NLB) *(pb_u32x4*)(dc + o) = __builtin_amdgcn_raw_buffer_load_b128(dr, o, 0, 0);
}

template <int GT> __global__ __launch_bounds__(kPbUpThreads) void k_pb_up_ws(PbUpArgs a)
{
	static_assert(GT == 1 || GT == 2, "value groups");
	static_assert(kPbWsNQ % kPbWsCh == 0, "whole chunks");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // at LDS address 0: a template word is the address
	uint8_t* const dcode0 = (uint8_t*)(win + a.pitch + kPbZeroSlots); // [2][pitch]
	int32_t* off_s = (int32_t*)(lds_raw + pb_ws_meta_offset(a.pitch));
	uint16_t* len_s = (uint16_t*)(off_s + a.spb * a.G);
	double* dict_s = (double*)(lds_raw + pb_ws_dict_offset(a.pitch, a.spb, a.G));
	for (int i = threadIdx.x; i < 256; i += kPbUpThreads) dict_s[i] = a.dict[i];
	for (int i = threadIdx.x; i < a.spb * a.G; i += kPbUpThreads) {
		off_s[i] = a.tw_off[i];
		len_s[i] = a.tw_len[i];
	}
	if (threadIdx.x < kPbZeroSlots) win[a.pitch + threadIdx.x] = 0.0; // never overwritten
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	double gco = 0.0;
	if (a.g_a) {
		gco = *a.g_a;
		const double b2 = *a.g_b2;
		if (sqrt(b2) >= 1e-10) gco /= b2;
	}
	constexpr int NG = kPbWsGather;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const bool loader = wave >= NG; // wave-uniform
	const int ll = (int)threadIdx.x - NG * 64;
	const int p2 = (int)(a.pitch >> 1);
	const int pitch = (int)a.pitch;
	const uint2* const tw2 = (const uint2*)a.tw;
	double* const uout = a.u;
	double* const wbuf = a.wbuf;
	const double* const ybuf = a.ybuf;
	const uint8_t* const dcode = a.dcode;
	const int spb = a.spb, n_up = (int)a.n_up;
	const int64_t n_blk = a.n_blk;
	const bool beta_in_u = a.beta_in_u != 0;
	double gv[GT];
#pragma unroll
	for (int g = 0; g < GT; g++) gv[g] = a.gval[g];
	auto gather4 = [=](const uint2& w, double& s0, double& s1) __attribute__((always_inline)) {
		s0 += pb_lds_abs(pb_lo8(w.x));
		s1 += pb_lds_abs(pb_hi8(w.x));
		s0 += pb_lds_abs(pb_lo8(w.y));
		s1 += pb_lds_abs(pb_hi8(w.y));
	};
	// Two loops with the same barrier sequence, one per role (the branch is wave-uniform).  Written as ONE loop with the roles
	// branching inside it, the loaders' 84 row registers stay allocated through the gather code: 344 spilled registers.
	const int64_t step = gridDim.x;
	if (loader) {
		double2 rr[kPbWsNQ];
		int cur = 0;
		int64_t blk = blockIdx.x;
		if (blk < n_blk) pb_ws_fetch(rr, wbuf + blk * pitch, (const double*)ybuf + blk * pitch, dcode + blk * pitch, dcode0, ll, pitch, gco);
		for (; blk < n_blk; blk += step) {
			__syncthreads(); // B1: the gather waves are done with the window (first block: the metadata is in place)
#pragma unroll
			for (int q = 0; q < kPbWsNQ; q++)
				if (ll + q * kPbWsLoaders * 64 < p2) ((double2*)win)[ll + q * kPbWsLoaders * 64] = rr[q];
			__syncthreads(); // B2: the window holds r_next of this block, the code buffer `cur` its diagonal codes
			const int64_t nb = blk + step;
			if (nb < n_blk) pb_ws_fetch(rr, wbuf + nb * pitch, (const double*)ybuf + nb * pitch, dcode + nb * pitch, dcode0 + (size_t)(cur ^ 1) * pitch, ll, pitch, gco);
			cur ^= 1;
		}
		return;
	}
	int cur = 0;
	for (int64_t blk = blockIdx.x; blk < n_blk; blk += step) {
		const int64_t rowbase = blk * pitch;
		__syncthreads(); // B1
		__syncthreads(); // B2
		const uint8_t* const dcode_s = dcode0 + (size_t)cur * pitch;
		const double* const yold = ybuf + rowbase;
		auto epilogue = [=](int j, double acc, double yo) __attribute__((always_inline)) {
			const int iu_raw = j * 64 + lane;
			const bool valid = iu_raw < n_up;
			const int iu = valid ? iu_raw : n_up - 1;
			acc = fma(dict_s[dcode_s[iu]], win[iu], acc);
			if (valid) __builtin_nontemporal_store(fma(beta, yo, alpha * acc), &uout[rowbase + iu]); // as k_pb_up<CHAIN>
		};
		auto load_words = [=](int j, PbWords<GT>& s) __attribute__((always_inline)) {
			if (j >= spb) return; // wave-uniform
#pragma unroll
			for (int g = 0; g < GT; g++) {
				s.nc[g] = len_s[j * GT + g];
				const uint2* wp = tw2 + (size_t)off_s[j * GT + g] * 64 + lane;
#pragma unroll
				for (int c = 0; c < kPbPre; c++) s.w[g][c] = wp[c * 64];
			}
			s.yo = beta_in_u ? yold[min(j * 64 + lane, n_up - 1)] : 0.0; // wave-uniform
		};
		auto group_sum = [=](int j, int g, int nc, const uint2* w) __attribute__((always_inline)) {
			double s0 = 0.0, s1 = 0.0;
			if (nc >= 2) {
				gather4(w[0], s0, s1);
				gather4(w[1], s0, s1);
			} else if (nc == 1) {
				gather4(w[0], s0, s1);
			}
			if (nc >= 4) {
				gather4(w[2], s0, s1);
				gather4(w[3], s0, s1);
			} else if (nc == 3) {
				gather4(w[2], s0, s1);
			}
			if (nc > kPbPre) {
				const uint2* wp = tw2 + (size_t)off_s[j * GT + g] * 64 + lane;
				for (int c = kPbPre; c < nc; c++) {
					const uint2 wr = wp[c * 64];
					gather4(wr, s0, s1);
				}
			}
			return s0 + s1;
		};
		auto compute = [=](int j, const PbWords<GT>& s) __attribute__((always_inline)) {
			if (j >= spb) return; // wave-uniform
			double acc = 0.0;
#pragma unroll
			for (int g = 0; g < GT; g++) acc = fma(gv[g], group_sum(j, g, s.nc[g], s.w[g]), acc);
			epilogue(j, acc, s.yo);
		};
		PbWords<GT> wa, wb;
		load_words(wave, wa);
		for (int j0 = wave; j0 < spb; j0 += 2 * NG) {
			load_words(j0 + NG, wb);
			compute(j0, wa);
			load_words(j0 + 2 * NG, wa);
			compute(j0 + NG, wb);
		}
		cur ^= 1;
	}
}

} // namespace lpp
