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
constexpr int kPbDownThreads = 512;

// ---------------------------------------------------------------------------------------------
// in-block part + diagonal:  x[b][i] = beta' x[b][i] + alpha ( sum_k T[i][c_k] y[b][c_k] + D[b][i] y[b][i] )   (+ Re<y|x> partial)
// ---------------------------------------------------------------------------------------------
struct PbUpArgs {
	// template: for slice j and value group g, npairs = tw_len[j*G+g] slot pairs at tw + tw_off[j*G+g]; word (pair p, lane l) at
	// [p*64 + l] holds two 16-bit LDS window indices (slots 2p and 2p+1); padding entries index a zero slot
	const uint32_t* tw;
	const int32_t* tw_off;
	const uint16_t* tw_len;
	int G;
	double gval[kPbMaxGroups];
	const double* dict; // 256 doubles (diagonal codes)
	const uint8_t* dcode; // one code per row, pitched like the vectors (null: no diagonal)
	int64_t n_up, pitch, n_blk;
	int spb; // slices per block
	const double* y;
	double* x;
	double* partial;
	EpiScale sc;
};

template <bool DOT> __global__ __launch_bounds__(kPbUpThreads) void k_pb_up(PbUpArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	double* win = (double*)lds_raw; // pitch + kPbZeroSlots elements
	__shared__ double smem[kPbUpThreads / 64];
	__shared__ double dict_s[256];
	__shared__ int next_slice;
	for (int i = threadIdx.x; i < 256; i += kPbUpThreads) dict_s[i] = a.dict[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	const int lane = threadIdx.x & 63;
	const int64_t p2 = a.pitch >> 1; // pitch is a multiple of 16
	double dot = 0.0;
	for (int64_t blk = blockIdx.x; blk < a.n_blk; blk += gridDim.x) {
		const double2* yb = (const double2*)(a.y + blk * a.pitch);
		__syncthreads(); // everyone is done with the previous window
		if (threadIdx.x == 0) next_slice = 0;
		for (int64_t i0 = threadIdx.x; i0 < p2; i0 += 8 * kPbUpThreads) {
			double2 t[8];
#pragma unroll
			for (int q = 0; q < 8; q++) t[q] = yb[min(i0 + (int64_t)q * kPbUpThreads, p2 - 1)];
#pragma unroll
			for (int q = 0; q < 8; q++)
				if (i0 + (int64_t)q * kPbUpThreads < p2) ((double2*)win)[i0 + (int64_t)q * kPbUpThreads] = t[q];
		}
		if (threadIdx.x < kPbZeroSlots) win[a.pitch + threadIdx.x] = 0.0;
		__syncthreads();
		const double* xb = a.x + blk * a.pitch;
		for (int j = next_slice_claim(&next_slice); j < a.spb; j = next_slice_claim(&next_slice)) {
			const int iu_raw = j * 64 + lane;
			const bool valid = iu_raw < a.n_up;
			const int iu = valid ? iu_raw : (int)a.n_up - 1;
			const double xold = xb[iu];
			uint32_t dc = 0;
			if (a.dcode) dc = a.dcode[blk * a.pitch + iu];
			double acc = 0.0;
			for (int g = 0; g < a.G; g++) { // wave-uniform trip counts
				const int np = a.tw_len[j * a.G + g];
				const uint32_t* wp = a.tw + a.tw_off[j * a.G + g] + lane;
				double s0 = 0.0, s1 = 0.0;
				uint32_t w0[4], w1[4];
				const int np4 = np & ~3;
				if (np4 > 0) {
#pragma unroll
					for (int q = 0; q < 4; q++) w0[q] = wp[q * 64];
				}
				for (int p = 0; p < np4; p += 4) { // word loads of the next four pairs are in flight behind these LDS gathers
					if (p + 4 < np4) {
#pragma unroll
						for (int q = 0; q < 4; q++) w1[q] = wp[(p + 4 + q) * 64];
					}
#pragma unroll
					for (int q = 0; q < 4; q++) {
						s0 += win[w0[q] & 0xffffu];
						s1 += win[w0[q] >> 16];
					}
#pragma unroll
					for (int q = 0; q < 4; q++) w0[q] = w1[q];
				}
				if (np4 < np) { // up to three pairs left: loaded together (clamped), the unused ones are skipped
					uint32_t wr[3];
#pragma unroll
					for (int q = 0; q < 3; q++) wr[q] = wp[min(np4 + q, np - 1) * 64];
#pragma unroll
					for (int q = 0; q < 3; q++) {
						if (np4 + q < np) {
							s0 += win[wr[q] & 0xffffu];
							s1 += win[wr[q] >> 16];
						}
					}
				}
				acc = fma(a.gval[g], s0 + s1, acc);
			}
			const double yc = win[iu];
			if (a.dcode) acc = fma(dict_s[dc], yc, acc);
			if (valid) {
				const double xv = epi_lin(beta, xold, alpha, acc);
				a.x[blk * a.pitch + iu] = xv;
				if (DOT) dot += yc * xv;
			}
		}
	}
	if (DOT) {
		const double r = block_sum_n<kPbUpThreads / 64>(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// block couplings:  x[b][i] = beta x[b][i] + alpha sum_k C[b][b'_k] y[b'_k][i]
// One persistent workgroup per CU.  Workgroup w belongs to group w mod 8 (one XCD under round-robin dispatch: speed only) and
// owns a fixed range of blocks for ALL panels of its group; the couplings of its blocks sit in LDS (byte offsets of the source
// blocks + value codes), so nothing but y and x moves through L2.  Group k walks panels k, k+8, ...; the workgroups of a group
// stay within two panels of each other (bounded pacing: per-group, per-panel counters), which keeps the panel in that L2.
// A wave covers 8 blocks x 16 positions with 16-byte lanes; x is streamed with non-temporal accesses.
// ---------------------------------------------------------------------------------------------
struct PbDownArgs {
	int64_t pitch, n_blk;
	int npanels; // pitch / 16
	int ids_per_wg; // blocks owned by one workgroup
	int rowcap; // LDS places per block (longest coupling list rounded up to a multiple of 8)
	const int64_t* c_ptr; // couplings: CSR over blocks, off-diagonal, ascending
	const int32_t* c_col;
	const uint8_t* c_code; // dictionary code of each coupling
	const double* dict;
	const double* y; // addressed with 32-bit byte offsets (< 4 GiB)
	double* x;
	EpiScale sc;
	int* pace; // [8][npanels] finished-workgroup counters (zeroed before the launch); null: free-running
	int nwaves; // waves per workgroup actually working (<= kPbDownThreads/64)
};

static __global__ __launch_bounds__(kPbDownThreads) void k_pb_down(PbDownArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	__shared__ double dict_s[256];
	uint32_t* off_s = (uint32_t*)lds_raw; // [ids_per_wg][rowcap] byte offset of the source block
	uint8_t* code_s = (uint8_t*)(off_s + (size_t)a.ids_per_wg * a.rowcap); // [ids_per_wg][rowcap]
	for (int i = threadIdx.x; i < 256; i += kPbDownThreads) dict_s[i] = a.dict[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 3, c = lane & 7;
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1; // groups the panels are dealt over
	const int grp = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int slot = nx == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
	const int nslots = (int)(gridDim.x / nx);
	const int64_t b0 = (int64_t)slot * a.ids_per_wg;
	const int nown = (int)max((int64_t)0, min((int64_t)a.ids_per_wg, a.n_blk - b0));
	const uint32_t rowbytes = (uint32_t)(a.pitch * 8);
	for (int i = threadIdx.x; i < nown * a.rowcap; i += kPbDownThreads) {
		const int il = i / a.rowcap, k = i - il * a.rowcap;
		const int64_t b = b0 + il;
		const int64_t p0 = a.c_ptr[b];
		const bool in = k < (int)(a.c_ptr[b + 1] - p0);
		// places beyond the list: the block itself with code 0 (+0.0): fixed trip count, no per-lane conditions
		off_s[i] = (uint32_t)(in ? a.c_col[p0 + k] : (int32_t)b) * rowbytes;
		code_s[i] = in ? a.c_code[p0 + k] : (uint8_t)0;
	}
	__syncthreads();
	const int ngroups = (nown + 7) >> 3;
	const int nchunk = a.rowcap >> 3;
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
		const uint32_t colb = (uint32_t)(p * 128 + c * 16); // byte offset of this lane's two positions inside a row
		if (wave < a.nwaves) {
			for (int g = wave; g < ngroups; g += a.nwaves) {
				const int il_raw = g * 8 + sub;
				const bool valid = il_raw < nown;
				const int il = min(il_raw, nown - 1);
				const size_t rowb = (size_t)(b0 + il) * rowbytes + colb;
				double2* xp = (double2*)((char*)a.x + rowb);
				const double2 xold = double2 { __builtin_nontemporal_load(&xp->x), __builtin_nontemporal_load(&xp->y) };
				const uint32_t* orow = off_s + il * a.rowcap;
				const uint8_t* crow = code_s + il * a.rowcap;
				double2 acc = double2 { 0.0, 0.0 };
				double2 g0[8], g1[8];
#pragma unroll
				for (int q = 0; q < 8; q++) g0[q] = *(const double2*)(ysrc + (size_t)(orow[q] + colb));
				for (int ch = 0; ch < nchunk; ch++) {
					if (ch + 1 < nchunk) {
#pragma unroll
						for (int q = 0; q < 8; q++) g1[q] = *(const double2*)(ysrc + (size_t)(orow[(ch + 1) * 8 + q] + colb));
					}
#pragma unroll
					for (int q = 0; q < 8; q++) {
						const double v = dict_s[crow[ch * 8 + q]];
						acc.x = fma(v, g0[q].x, acc.x);
						acc.y = fma(v, g0[q].y, acc.y);
					}
#pragma unroll
					for (int q = 0; q < 8; q++) g0[q] = g1[q];
				}
				if (valid) {
					const double2 xv = double2 { beta * xold.x + alpha * acc.x, beta * xold.y + alpha * acc.y };
					__builtin_nontemporal_store(xv.x, &xp->x);
					__builtin_nontemporal_store(xv.y, &xp->y);
				}
			}
		}
		if (a.pace) {
			__syncthreads();
			if (threadIdx.x == 0) __hip_atomic_fetch_add(a.pace + (int64_t)grp * a.npanels + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
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
                                    double* __restrict__ val_out)
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
	val_out[o] = dict[dcode[b * pitch + i]];
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

// start vector in the pitched layout: element (b, i) takes the value the unpitched stream gives index b*rows + i
static __global__ void k_fill_random_pitched(double* __restrict__ v, int64_t n_blk, int64_t rows, int64_t pitch, int64_t offset, uint64_t seed)
{
	const int64_t n = n_blk * pitch;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = k / pitch, i = k - b * pitch;
		double val = 0.0;
		if (i < rows) {
			const uint64_t r = splitmix64(seed * 0x2545F4914F6CDD1DULL + (uint64_t)(b * rows + i + offset));
			val = (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5;
		}
		v[k] = val;
	}
}

} // namespace lpp
