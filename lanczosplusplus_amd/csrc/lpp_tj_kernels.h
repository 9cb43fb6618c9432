// lpp_tj_kernels.h -- the one-orbital t-J Hamiltonian (TjMultiOrb.h:100-131, 586-783) WITHOUT a stored matrix, in a HOLE-MAJOR order
// of the basis (round 5; BASELINE config 4).
//
// The reference's basis (BasisTjMultiOrbLanczos.h:29-42, 354-369) is the sorted list of words (down << L) | up without double
// occupancy: neither a product basis nor local -- a hop of a hole moves a row by hundreds of thousands of positions, and the
// general layout streams 1.7 GB of (column, value code) pairs per product beside 2.4 GB of gathers that miss L2.
// A state is equally well (H, sigma): H = the set of holes, sigma = the spin pattern of the Lo = nup + ndown occupied sites read
// in site order (bit k of sigma: the k-th occupied site holds an up electron).  Stored position = block(H) * pitch + rank(sigma),
// every block holding the same C(Lo, nup) patterns in ascending order.  In these coordinates (derived from the reference's
// element formulas, cited at each term):
//   S+S-  (TjMultiOrb.h:697-783) on a bond whose two sites are occupied flips an antiparallel pair of sigma at the COMPRESSED
//         positions p < q: column = (H, sigma ^ (bit p | bit q)).  signSplusSminus counts the electrons on the sites [i, j) of the
//         bra: site i and every occupied site between, i.e. q - p of them -- a constant of (H, bond).  Value 0.5 J (-1)^(q-p).
//   hop   (TjMultiOrb.h:649-695) moves the electron at position p (either spin) onto a neighbouring hole: the block changes, the
//         m electrons between the two sites shift by one position (a rotation of the bits [lo, lo + m] of sigma), value
//         h (-1)^(electrons of the SAME species between) -- extraSign and doSign together (:674-692); h = hoppings(min, max)
//         unconjugated for both directions, as the reference has it.
//   diag  (TjMultiOrb.h:586-647) one f64 per row, formed once from the state with the assembler's diag_of (bit for bit the reference's sum).
// Nothing per entry is stored: a block is described by its bonds (<= 96 x 16 bytes) and its hops (<= 64 x 24 bytes), the ranking of
// a pattern is two LDS table reads (high half: number of patterns below it, low half: rank inside its popcount class).
// Per product the kernel reads the vector once from memory, gathers from L2 (the block in flight and its hop neighbours) and the
// Infinity Cache, and writes it once.  Only the boundary knows the order (vec_from_host / _to_host / the start vector:
// TjState::perm); lpp_engine_get_csr re-runs the device assembler in the reference's order.
#pragma once
#include "lpp_kernels.h"
#include "lpp_tj.h"

namespace lpp {

struct TjArgs {
	const uint32_t* pat; // [ns] spin patterns, ascending
	const int32_t* hi_base; // [1 << hb]
	const uint16_t* lo_rank; // [1 << lb]
	int lb, nhi, nlo;
	int ns; // patterns per block
	int64_t pitch; // elements between blocks (> ns: element ns of every block is zero)
	int nblk;
	int chunks; // work items per block
	const TjItem* items; // [chunks], the same for every block
	const TjBlock* blocks;
	const TjPair* pairs;
	const TjHop* hops;
	const int32_t* order; // [nblk] blocks in processing order, XCD x takes the x-th eighth
	const double* diag; // [nblk * pitch]
	const void* y;
	void* x;
	const void* ydot; // null: no partials
	double* partial;
	EpiScale sc;
};

// dynamic LDS of k_tj_apply: window (kTjWindow + 1 elements) | hi_base[nhi] | lo_rank[nlo]
__host__ __device__ inline size_t tj_lds_bytes(size_t esz, int nhi, int nlo) { return esz * (size_t)(kTjWindow + 1) + sizeof(int32_t) * (size_t)nhi + ((sizeof(uint16_t) * (size_t)nlo + 15) & ~(size_t)15); }

template <typename T> __device__ __forceinline__ void tj_mac_real(T& acc, double v, const T& y);
template <> __device__ __forceinline__ void tj_mac_real<double>(double& acc, double v, const double& y) { acc = fma(v, y, acc); }
template <> __device__ __forceinline__ void tj_mac_real<cplx>(cplx& acc, double v, const cplx& y)
{
	acc.re = fma(v, y.re, acc.re);
	acc.im = fma(v, y.im, acc.im);
}

// x = beta x + alpha H y  (+ partials of Re<ydot|x>).  One work item = a run of <= kTjWindow consecutive patterns of one block made of
// whole segments; its piece of y is staged in LDS.  A thread takes kTjRowsPerThread rows per round (lanes = consecutive patterns: a flip
// or a rotation among the high positions moves all 64 by the same amount -- coalesced --, one among the low positions keeps them inside
// a few lines).  The bond / hop loop is the outer loop: its scalars are read from LDS once for a thread's rows, whose gathers are in
// flight together.  Flips among the low positions are LDS reads (54 % of the bonds of the 4x5 lattice with two holes): the first version
// gathered everything from memory and had the CU's vector-memory pipeline busy 82 % of the time with 48 wave-level loads per 64 rows.
// CH: complex hopping amplitudes (T = cplx only).
template <typename T, bool CH, bool DOT> __global__ __launch_bounds__(kTjThreads) void k_tj_apply(TjArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char tj_lds[];
	T* const win = (T*)tj_lds; // [kTjWindow + 1]: the item's run of y, then one zero element
	int32_t* const hi_s = (int32_t*)(win + kTjWindow + 1);
	uint16_t* const lo_s = (uint16_t*)(hi_s + a.nhi);
	__shared__ TjPair pairs_s[kTjMaxPairs];
	__shared__ TjHop hops_s[kTjMaxHops];
	__shared__ double smem[kTjThreads / 64];
	for (int i = threadIdx.x; i < a.nhi; i += kTjThreads) hi_s[i] = a.hi_base[i];
	for (int i = threadIdx.x; i < a.nlo; i += kTjThreads) lo_s[i] = a.lo_rank[i];
	double alpha, beta;
	epi_coeffs(a.sc, alpha, beta);
	const int lb = a.lb;
	const uint32_t lbm = (1u << lb) - 1u;
	const int ns = a.ns;
	const T* const yv = (const T*)a.y;
	T* const xv = (T*)a.x;
	const T* const yd = (const T*)a.ydot;
	double dot = 0.0;
	// blocks of this XCD (round-robin dispatch: workgroup b runs on XCD b mod 8; speed only), walked together by its workgroups
	const int nx = (gridDim.x & 7) == 0 ? 8 : 1;
	const int xcd = nx == 8 ? (int)(blockIdx.x & 7) : 0;
	const int slot = nx == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, nslots = (int)gridDim.x / nx;
	const int k0 = (int)((int64_t)a.nblk * xcd / nx), k1 = (int)((int64_t)a.nblk * (xcd + 1) / nx);
	const int64_t total = (int64_t)(k1 - k0) * a.chunks;
	auto rank_of = [&](uint32_t s) __attribute__((always_inline)) -> int { return hi_s[s >> lb] + (int)lo_s[s & lbm]; };
	for (int64_t seq = slot; seq < total; seq += nslots) {
		// block-major: the XCD's workgroups hold ~3 consecutive blocks at a time, whose flips among the high positions hit this L2.  (Item-major
		// inside the XCD's group of blocks -- the same few items of all its 24 blocks at a time, so that the pieces the blocks gather from each
		// other would be in L2 -- was measured: 2.27 instead of 1.93 GB of fabric reads, 0.373 instead of 0.361 ms.)
		const int blk = a.order[k0 + (int)(seq / a.chunks)];
		const TjItem it = a.items[(int)(seq % a.chunks)];
		const TjBlock B = a.blocks[blk];
		const int64_t rowbase = (int64_t)blk * a.pitch;
		const T* const yb = yv + rowbase;
		__syncthreads(); // the previous item is done with the window and the lists (and the rank tables are in place)
		if ((int)threadIdx.x < B.nx) pairs_s[threadIdx.x] = a.pairs[B.x_first + threadIdx.x];
		if ((int)threadIdx.x < B.nh) hops_s[threadIdx.x] = a.hops[B.h_first + threadIdx.x];
		for (int i = threadIdx.x; i < it.len; i += kTjThreads) win[i] = yb[it.r0 + i];
		if (threadIdx.x == 0) win[it.len] = VT<T>::zero();
		__syncthreads();
		for (int base = 0; base < it.len; base += kTjThreads * kTjRowsPerThread) { // (uniform trip count)
			uint32_t sg[kTjRowsPerThread];
			int rl[kTjRowsPerThread]; // row inside the item
			T acc[kTjRowsPerThread];
#pragma unroll
			for (int k = 0; k < kTjRowsPerThread; k++) {
				rl[k] = base + k * kTjThreads + (int)threadIdx.x;
				sg[k] = a.pat[it.r0 + min(rl[k], it.len - 1)];
				acc[k] = VT<T>::zero();
			}
			// S+S- among the low positions: the flipped pattern lies in the same segment -- an LDS read; parallel pairs read the zero element
			for (int e = 0; e < B.nxl; e++) {
				const uint32_t mask = pairs_s[e].mask;
				const double v = pairs_s[e].v;
				T g[kTjRowsPerThread];
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) {
					const bool anti = __popc(sg[k] & mask) == 1;
					const int rr = rank_of(sg[k] ^ mask) - it.r0;
					g[k] = win[anti ? rr : it.len];
				}
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) tj_mac_real<T>(acc[k], v, g[k]);
			}
			// the other bonds: from the block's row in memory (32-bit byte offset against the block's uniform base)
			for (int e = B.nxl; e < B.nx; e++) {
				const uint32_t mask = pairs_s[e].mask;
				const double v = pairs_s[e].v;
				T g[kTjRowsPerThread];
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) {
					const bool anti = __popc(sg[k] & mask) == 1;
					const uint32_t rr = (uint32_t)rank_of(sg[k] ^ mask);
					g[k] = *(const T*)((const char*)yb + (anti ? rr : (uint32_t)ns) * (uint32_t)sizeof(T));
				}
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) tj_mac_real<T>(acc[k], v, g[k]);
			}
			// hops: the electron at one end of the bit range moves to the other end, the electrons between shift by one
			for (int h = 0; h < B.nh; h++) {
				const TjHop hp = hops_s[h];
				const int lo = hp.lo, m = hp.m;
				const uint32_t wm = (2u << m) - 1u; // m + 1 bits
				const T* const ys = yv + (int64_t)hp.dst * a.pitch;
				T g[kTjRowsPerThread];
				uint32_t par[kTjRowsPerThread];
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) {
					const uint32_t seg = (sg[k] >> lo) & wm;
					uint32_t b, nseg, ups;
					if (hp.dir == 0) { // wave-uniform
						b = seg & 1u;
						nseg = (seg >> 1) | (b << m);
						ups = (uint32_t)__popc(seg >> 1);
					} else {
						b = (seg >> m) & 1u;
						nseg = ((seg << 1) & wm) | b;
						ups = (uint32_t)__popc(seg & (wm >> 1));
					}
					// an up electron passes the up electrons between, a down electron the down electrons (m - ups of them)
					par[k] = (b ? ups : (uint32_t)m - ups) & 1u;
					const uint32_t s2 = (sg[k] & ~(wm << lo)) | (nseg << lo);
					g[k] = *(const T*)((const char*)ys + (uint32_t)min(rank_of(s2), ns - 1) * (uint32_t)sizeof(T)); // (non-temporal hop reads: 0.406 instead of 0.356 ms)
				}
#pragma unroll
				for (int k = 0; k < kTjRowsPerThread; k++) {
					const double vr = par[k] ? -hp.vr : hp.vr;
					if constexpr (CH) {
						const double vi = par[k] ? -hp.vi : hp.vi;
						VT<T>::mac(acc[k], T { vr, vi }, g[k]);
					} else
						tj_mac_real<T>(acc[k], vr, g[k]);
				}
			}
#pragma unroll
			for (int k = 0; k < kTjRowsPerThread; k++) {
				if (rl[k] >= it.len) continue;
				const int64_t at = rowbase + it.r0 + rl[k];
				tj_mac_real<T>(acc[k], a.diag[at], win[rl[k]]);
				const T xn = epi_lin(beta, xv[at], alpha, acc[k]);
				xv[at] = xn;
				if (DOT) dot += VT<T>::dot_re(yd[at], xn);
			}
		}
	}
	if (DOT) {
		const double s = block_sum(dot, smem);
		if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
	}
}

// ---- one-off kernels of the layout ----------------------------------------------------------------------------------------------

// software pdep over the occupied sites: bit k of v goes to the k-th set bit of mask
__device__ __forceinline__ uint64_t tj_pdep(uint64_t v, uint64_t mask)
{
	uint64_t out = 0;
	while (mask) {
		const uint64_t low = mask & (~mask + 1);
		if (v & 1) out |= low;
		v >>= 1;
		mask &= mask - 1;
	}
	return out;
}

} // namespace lpp
