// lpp_tj.hip -- host side of the hole-major, matrix-free form of the one-orbital t-J Hamiltonian (kernel and rationale: lpp_tj_kernels.h).
// Citations are relative to /root/reference/src.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lpp_assemble_kernels.h"
#include "lpp_engine_impl.h"
#include "lpp_tj_kernels.h"

using namespace lpp;

namespace {

struct Buf {
	void* p = nullptr;
	~Buf()
	{
		if (p) (void)hipFree(p);
	}
};

template <typename V> lpp_status to_dev(V** dst, const std::vector<V>& src, hipStream_t st)
{
	HIP_TRY_MEM(hipMalloc((void**)dst, sizeof(V) * std::max<size_t>(src.size(), 1)));
	if (!src.empty()) HIP_TRY(hipMemcpyAsync(*dst, src.data(), sizeof(V) * src.size(), hipMemcpyHostToDevice, st));
	return LPP_OK;
}

uint64_t binom_h(int n, int k)
{
	if (k < 0 || k > n) return 0;
	uint64_t r = 1;
	for (int i = 1; i <= k; i++) r = r * (uint64_t)(n - k + i) / (uint64_t)i;
	return r;
}

// all nbits-bit words with k set bits, ascending (the loop of BasisOneSpin.h:53-61 / BasisTjMultiOrbLanczos.h:323-352)
std::vector<uint32_t> words_of(int nbits, int k)
{
	std::vector<uint32_t> out;
	if (k == 0) {
		out.push_back(0);
		return out;
	}
	if (k > nbits) return out;
	uint32_t w = (k >= 32) ? ~0u : ((1u << k) - 1u);
	const uint64_t limit = 1ull << nbits;
	while ((uint64_t)w < limit) {
		out.push_back(w);
		const uint32_t c = w & (0u - w), r = w + c; // Gosper's next word with the same popcount
		if (r == 0) break;
		w = (((r ^ w) >> 2) / c) | r;
	}
	return out;
}

// stored (block, pattern) -> the reference's basis index and diagonal element.  The state word is put together from the hole set and the
// spin pattern (bit k of the pattern: the k-th occupied site holds an up electron); index and diagonal are the device assembler's
// (index_of / diag_of: BasisTjMultiOrbLanczos.h:29-42, TjMultiOrb.h:586-647).
__global__ __launch_bounds__(kBlock) void k_tj_perm_diag(AsmParams P, const uint32_t* __restrict__ holes, const uint32_t* __restrict__ pat, int nblk, int ns, int Lo,
                                                         int64_t pitch, int32_t* __restrict__ perm, double* __restrict__ diag)
{
	const int64_t n = (int64_t)nblk * ns;
	const uint64_t lowmask = (1ull << P.L) - 1;
	for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
		const int blk = (int)(k / ns), r = (int)(k - (int64_t)blk * ns);
		const uint64_t occ = ~(uint64_t)holes[blk] & lowmask;
		const uint64_t sg = pat[r];
		const uint64_t up = tj_pdep(sg, occ), down = tj_pdep(~sg & ((1ull << Lo) - 1), occ);
		const uint64_t w = (down << P.L) | up;
		perm[k] = (int32_t)index_of<ASM_TJ>(P, w);
		diag[(int64_t)blk * pitch + r] = diag_of<ASM_TJ>(P, w);
	}
}

// basis order (contiguous) <-> stored order (pitched); padding elements are left alone (zero)
template <typename T> __global__ void k_tj_gather(T* __restrict__ dst, const T* __restrict__ src, const int32_t* __restrict__ perm, int nblk, int ns, int64_t pitch)
{
	const int64_t n = (int64_t)nblk * ns;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int blk = (int)(k / ns), r = (int)(k - (int64_t)blk * ns);
		dst[(int64_t)blk * pitch + r] = src[perm[k]];
	}
}
template <typename T> __global__ void k_tj_scatter(T* __restrict__ dst, const T* __restrict__ src, const int32_t* __restrict__ perm, int nblk, int ns, int64_t pitch)
{
	const int64_t n = (int64_t)nblk * ns;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int blk = (int)(k / ns), r = (int)(k - (int64_t)blk * ns);
		dst[perm[k]] = src[(int64_t)blk * pitch + r];
	}
}
// the built-in start vector: the element at basis index i is the one the unpermuted stream gives index i (k_fill_random), bit for bit
__global__ void k_tj_fill_random(double* __restrict__ v, const int32_t* __restrict__ perm, int nblk, int ns, int64_t pitch, int comp, uint64_t seed)
{
	const int64_t n = (int64_t)nblk * ns * comp;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
		const int64_t el = k / comp;
		const int c = (int)(k - el * comp);
		const int blk = (int)(el / ns), r = (int)(el - (int64_t)blk * ns);
		const uint64_t q = splitmix64(seed * 0x2545F4914F6CDD1DULL + (uint64_t)((int64_t)perm[el] * comp + c));
		v[((int64_t)blk * pitch + r) * comp + c] = (double)(q >> 11) * (1.0 / 9007199254740992.0) - 0.5;
	}
}

__global__ void k_tj_sum_i64(const int64_t* __restrict__ v, int64_t n, unsigned long long* __restrict__ out)
{
	unsigned long long s = 0;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) s += (unsigned long long)v[k];
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// largest |a - b| and largest |b| over n doubles, as the bit patterns of non-negative doubles (ordered like unsigned integers)
__global__ void k_tj_max_diff(int64_t n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out)
{
	double d = 0.0, m = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		d = fmax(d, fabs(a[i] - b[i]));
		m = fmax(m, fabs(b[i]));
		if (a[i] != a[i]) d = 1e300;
	}
	atomicMax(out, (unsigned long long)__double_as_longlong(d));
	atomicMax(out + 1, (unsigned long long)__double_as_longlong(m));
}

int blocks_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1 << 16)); }

} // namespace

namespace lpp {

void free_tj(lpp_engine* e)
{
	TjState& S = e->tj;
	const bool was = S.active;
	for (void* p : { (void*)S.pat, (void*)S.hi_base, (void*)S.lo_rank, S.blocks, S.pairs, S.hops, S.items, (void*)S.order, (void*)S.diag, (void*)S.perm })
		if (p) (void)hipFree(p);
	S = TjState();
	if (was) e->pitch = e->pitch_rows = e->pitch_blocks = 0;
}

int tj_launch(lpp_engine* e, const void* src, void* x, const void* ydot, double* partial, const EpiScale& sc)
{
	const TjState& S = e->tj;
	TjArgs a {};
	a.pat = S.pat;
	a.hi_base = S.hi_base;
	a.lo_rank = S.lo_rank;
	a.lb = S.lb;
	a.nhi = S.nhi;
	a.nlo = S.nlo;
	a.ns = S.ns;
	a.pitch = S.pitch;
	a.nblk = S.nblk;
	a.chunks = S.chunks;
	a.items = (const TjItem*)S.items;
	a.blocks = (const TjBlock*)S.blocks;
	a.pairs = (const TjPair*)S.pairs;
	a.hops = (const TjHop*)S.hops;
	a.order = S.order;
	a.diag = S.diag;
	a.y = src;
	a.x = x;
	a.ydot = ydot;
	a.partial = partial;
	a.sc = sc;
	const bool dot = ydot != nullptr && partial != nullptr;
	hipStream_t st = e->stream;
	const size_t lds = tj_lds_bytes(e->esz, S.nhi, S.nlo);
	if (!e->is_complex) {
		if (dot)
			k_tj_apply<double, false, true><<<S.grid, kTjThreads, lds, st>>>(a);
		else
			k_tj_apply<double, false, false><<<S.grid, kTjThreads, lds, st>>>(a);
	} else if (S.cplx_hops) {
		if (dot)
			k_tj_apply<cplx, true, true><<<S.grid, kTjThreads, lds, st>>>(a);
		else
			k_tj_apply<cplx, true, false><<<S.grid, kTjThreads, lds, st>>>(a);
	} else {
		if (dot)
			k_tj_apply<cplx, false, true><<<S.grid, kTjThreads, lds, st>>>(a);
		else
			k_tj_apply<cplx, false, false><<<S.grid, kTjThreads, lds, st>>>(a);
	}
	return dot ? S.grid : 0;
}

lpp_status tj_vec_from_host(lpp_engine* e, double* dev, const void* host)
{
	const TjState& S = e->tj;
	Buf land;
	const size_t bytes = e->esz * (size_t)e->n_local;
	HIP_TRY_MEM(hipMalloc(&land.p, bytes));
	HIP_TRY(hipMemcpyAsync(land.p, host, bytes, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipMemsetAsync(dev, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
	const int nb = blocks_for((int64_t)S.nblk * S.ns);
	if (e->is_complex)
		k_tj_gather<cplx><<<nb, 256, 0, e->stream>>>((cplx*)dev, (const cplx*)land.p, S.perm, S.nblk, S.ns, S.pitch);
	else
		k_tj_gather<double><<<nb, 256, 0, e->stream>>>(dev, (const double*)land.p, S.perm, S.nblk, S.ns, S.pitch);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream)); // the landing buffer goes away with this call
	return LPP_OK;
}

lpp_status tj_vec_to_host(lpp_engine* e, void* host, const double* dev)
{
	const TjState& S = e->tj;
	Buf land;
	const size_t bytes = e->esz * (size_t)e->n_local;
	HIP_TRY_MEM(hipMalloc(&land.p, bytes));
	const int nb = blocks_for((int64_t)S.nblk * S.ns);
	if (e->is_complex)
		k_tj_scatter<cplx><<<nb, 256, 0, e->stream>>>((cplx*)land.p, (const cplx*)dev, S.perm, S.nblk, S.ns, S.pitch);
	else
		k_tj_scatter<double><<<nb, 256, 0, e->stream>>>((double*)land.p, dev, S.perm, S.nblk, S.ns, S.pitch);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(host, land.p, bytes, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return LPP_OK;
}

void tj_fill_random(lpp_engine* e, double* dev, uint64_t seed)
{
	const TjState& S = e->tj;
	(void)hipMemsetAsync(dev, 0, sizeof(double) * (size_t)e->nd_pad, e->stream);
	const int comp = e->is_complex ? 2 : 1;
	k_tj_fill_random<<<blocks_for((int64_t)S.nblk * S.ns * comp), 256, 0, e->stream>>>(dev, S.perm, S.nblk, S.ns, S.pitch, comp, seed);
}

// The layout from the model's parameters alone: nothing of the matrix is ever stored.  Before it is used, ONE product of a random
// vector goes through it and through the device assembler's row walk (k_asm_apply: the term list of lpp_engine_assemble_tj, every
// entry re-derived per row in the reference's order); the largest difference must be below 1e-12 of the largest element.
// whether the hole-major form would be taken for this model at all (switches, sizes): the part of tj_build that needs no device work
bool tj_applies(const lpp_engine* e, const TjModel& M)
{
	bool forced = false;
	if (const char* s = getenv("LPP_TJ_LAYOUT")) {
		if (atoi(s) == 0) return false;
		forced = true;
	}
	if (e->cfg.spmv_kernel != LPP_SPMV_AUTO || getenv("LPP_SPMV_KERNEL")) return false;
	for (const char* k : { "LPP_SHARED_OFFSETS", "LPP_LOCAL16", "LPP_DIAG_CODES", "LPP_BLOCK_TEMPLATE", "LPP_WINDOW_ROWS", "LPP_COMPRESS_VALUES", "LPP_KEEP_PLAIN_CSR" })
		if (getenv(k)) return false; // switches of the general layout: measure that one
	if (e->cfg.compress_values == 0) return false;
	const int L = M.L, nup = M.nup, ndown = M.ndown, Lo = nup + ndown, nholes = L - Lo;
	if (Lo < 2 || Lo > 2 * kTjMaxHalf || nup < 1 || ndown < 1 || L > 31 || nholes < 0) return false;
	const uint64_t ns64 = binom_h(Lo, nup), nblk64 = binom_h(L, nholes);
	if (ns64 * nblk64 >= ((uint64_t)1 << 31) || ns64 < 64 || nblk64 > (1u << 20)) return false;
	// from 32 MB per vector on (as the product-basis layout of the Hubbard matrices); below that the general layout's launch is shorter
	if (!forced && ns64 * nblk64 * e->esz < ((uint64_t)32 << 20)) return false;
	if (M.has_im && !e->is_complex) return false;
	return true;
}

lpp_status tj_build(lpp_engine* e, const TjModel& M, const AsmParams& P, bool* done)
{
	*done = false;
	const bool verbose = getenv("LPP_VERBOSE") != nullptr;
	if (!tj_applies(e, M)) return LPP_OK;
	const int L = M.L, nup = M.nup, ndown = M.ndown, Lo = nup + ndown, nholes = L - Lo;
	const uint64_t ns64 = binom_h(Lo, nup), nblk64 = binom_h(L, nholes);
	const int ns = (int)ns64, nblk = (int)nblk64;
	hipStream_t st = e->stream;
	// the plan (host, no device work: lpp_tj_host.cpp): spin patterns and their ranking, work items, every hole configuration's bonds and moves
	TjPlan PL;
	{
		bool planned = false;
		std::string why;
		tj_plan(M, PL, &planned, &why);
		if (!planned) {
			if (verbose) fprintf(stderr, "lpp: t-J hole-major form does not apply: %s\n", why.c_str());
			return LPP_OK;
		}
	}
	if (PL.ns != ns || PL.nblk != nblk) return fail(LPP_ERR_INVALID, "tj_build: plan of another size");
	const int lb = PL.lb, hb = PL.hb, kbits = PL.kbits;
	const std::vector<uint32_t>& pat = PL.pat;
	const std::vector<uint32_t>& holes = PL.holes;
	const std::vector<int32_t>& hi_base = PL.hi_base;
	const std::vector<uint16_t>& lo_rank = PL.lo_rank;
	const std::vector<TjItem>& items = PL.items;
	const std::vector<TjBlock>& blocks = PL.blocks;
	const std::vector<TjPair>& pairs = PL.pairs;
	const std::vector<TjHop>& hops = PL.hops;
	const bool cplx_hops = PL.cplx_hops;
	if (cplx_hops && !e->is_complex) return LPP_OK;
	// processing order of the blocks: ascending hole words (neighbouring configurations share most of their hop sources)
	std::vector<int32_t> order((size_t)nblk);
	for (int b = 0; b < nblk; b++) order[(size_t)b] = b;
	free_tj(e);
	TjState& S = e->tj;
	struct Undo { // until the check below has passed the engine must not describe this form
		lpp_engine* e;
		bool* done;
		~Undo()
		{
			if (!*done) free_tj(e);
		}
	} undo { e, done };
	S.model = M;
	S.Lo = Lo;
	S.lb = lb;
	S.nhi = 1 << hb;
	S.nlo = 1 << lb;
	S.ns = ns;
	S.nblk = nblk;
	S.chunks = (int)items.size();
	S.kbits = kbits;
	const int64_t line = e->is_complex ? 8 : 16; // elements per 128-byte line
	S.pitch = ((int64_t)ns + 1 + line - 1) / line * line; // > ns: element ns of every block stays zero (what a parallel pair reads)
	S.cplx_hops = cplx_hops;
	{
		const int64_t per_xcd = ((int64_t)nblk + 7) / 8 * S.chunks;
		const int nslots = (int)std::max<int64_t>(1, std::min<int64_t>(per_xcd, 6 * std::max(1, e->num_cus / 8)));
		S.grid = 8 * nslots;
	}
	lpp_status rc;
	if ((rc = to_dev(&S.pat, pat, st)) != LPP_OK) return rc;
	if ((rc = to_dev(&S.hi_base, hi_base, st)) != LPP_OK) return rc;
	if ((rc = to_dev(&S.lo_rank, lo_rank, st)) != LPP_OK) return rc;
	if ((rc = to_dev((TjBlock**)&S.blocks, blocks, st)) != LPP_OK) return rc;
	if ((rc = to_dev((TjPair**)&S.pairs, pairs, st)) != LPP_OK) return rc;
	if ((rc = to_dev((TjHop**)&S.hops, hops, st)) != LPP_OK) return rc;
	if ((rc = to_dev(&S.order, order, st)) != LPP_OK) return rc;
	if ((rc = to_dev((TjItem**)&S.items, items, st)) != LPP_OK) return rc;
	uint32_t* d_holes = nullptr;
	Buf holes_buf;
	if ((rc = to_dev(&d_holes, holes, st)) != LPP_OK) return rc;
	holes_buf.p = d_holes;
	const int64_t n = (int64_t)nblk * ns, nstored = (int64_t)nblk * S.pitch;
	HIP_TRY_MEM(hipMalloc((void**)&S.perm, sizeof(int32_t) * (size_t)n));
	HIP_TRY_MEM(hipMalloc((void**)&S.diag, sizeof(double) * (size_t)nstored));
	HIP_TRY(hipMemsetAsync(S.diag, 0, sizeof(double) * (size_t)nstored, st));
	k_tj_perm_diag<<<blocks_for(n), kBlock, 0, st>>>(P, d_holes, S.pat, nblk, ns, Lo, S.pitch, S.perm, S.diag);
	S.table_bytes = (int64_t)(sizeof(uint32_t) * pat.size() + sizeof(int32_t) * hi_base.size() + sizeof(uint16_t) * lo_rank.size() + sizeof(TjBlock) * blocks.size()
	                          + sizeof(TjPair) * pairs.size() + sizeof(TjHop) * hops.size() + sizeof(int32_t) * order.size() + sizeof(TjItem) * items.size());
	// ---- entries of the CSR this stands for (the assembler's counting pass) -------------------------------------------------------
	Buf d_len, d_sum;
	HIP_TRY_MEM(hipMalloc(&d_len.p, sizeof(int64_t) * (size_t)n));
	HIP_TRY_MEM(hipMalloc(&d_sum.p, sizeof(unsigned long long)));
	HIP_TRY(hipMemsetAsync(d_sum.p, 0, sizeof(unsigned long long), st));
	k_asm_count<ASM_TJ><<<blocks_for(n), kBlock, 0, st>>>(P, (int64_t*)d_len.p);
	k_tj_sum_i64<<<1024, 256, 0, st>>>((const int64_t*)d_len.p, n, (unsigned long long*)d_sum.p);
	unsigned long long nnz = 0;
	HIP_TRY(hipMemcpyAsync(&nnz, d_sum.p, sizeof(nnz), hipMemcpyDeviceToHost, st));
	// ---- the check: one product through this form and through the assembler's row walk --------------------------------------------
	const int comp = e->is_complex ? 2 : 1;
	const size_t vb = e->esz * (size_t)n, vs = e->esz * (size_t)nstored;
	Buf d_y, d_xr, d_ys, d_xs, d_cmp;
	HIP_TRY_MEM(hipMalloc(&d_y.p, vb));
	HIP_TRY_MEM(hipMalloc(&d_xr.p, vb));
	HIP_TRY_MEM(hipMalloc(&d_ys.p, vs));
	HIP_TRY_MEM(hipMalloc(&d_xs.p, vs));
	HIP_TRY_MEM(hipMalloc(&d_cmp.p, sizeof(unsigned long long) * 2));
	HIP_TRY(hipMemsetAsync(d_xr.p, 0, vb, st));
	HIP_TRY(hipMemsetAsync(d_ys.p, 0, vs, st));
	HIP_TRY(hipMemsetAsync(d_xs.p, 0, vs, st));
	HIP_TRY(hipMemsetAsync(d_cmp.p, 0, sizeof(unsigned long long) * 2, st));
	k_fill_random<<<1024, 256, 0, st>>>((double*)d_y.p, n * comp, 0, 4711);
	const int nbr = blocks_for(n);
	const EpiScale one { nullptr, nullptr, 0 };
	e->pitch = S.pitch; // (tj_launch reads the state only)
	if (e->is_complex) {
		k_asm_apply<ASM_TJ, cplx, false><<<nbr, kBlock, 0, st>>>(P, (const cplx*)d_y.p, (cplx*)d_xr.p, nullptr, one);
		k_tj_gather<cplx><<<nbr, 256, 0, st>>>((cplx*)d_ys.p, (const cplx*)d_y.p, S.perm, nblk, ns, S.pitch);
	} else {
		k_asm_apply<ASM_TJ, double, false><<<nbr, kBlock, 0, st>>>(P, (const double*)d_y.p, (double*)d_xr.p, nullptr, one);
		k_tj_gather<double><<<nbr, 256, 0, st>>>((double*)d_ys.p, (const double*)d_y.p, S.perm, nblk, ns, S.pitch);
	}
	tj_launch(e, d_ys.p, d_xs.p, nullptr, nullptr, one);
	if (e->is_complex)
		k_tj_scatter<cplx><<<nbr, 256, 0, st>>>((cplx*)d_y.p, (const cplx*)d_xs.p, S.perm, nblk, ns, S.pitch); // back into the basis order (d_y is free now)
	else
		k_tj_scatter<double><<<nbr, 256, 0, st>>>((double*)d_y.p, (const double*)d_xs.p, S.perm, nblk, ns, S.pitch);
	k_tj_max_diff<<<1024, 256, 0, st>>>(n * comp, (const double*)d_y.p, (const double*)d_xr.p, (unsigned long long*)d_cmp.p);
	unsigned long long cmp[2] = { 0, 0 };
	HIP_TRY(hipMemcpyAsync(cmp, d_cmp.p, sizeof(cmp), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	e->pitch = 0;
	double dmax, xmax;
	std::memcpy(&dmax, &cmp[0], 8);
	std::memcpy(&xmax, &cmp[1], 8);
	if (verbose)
		fprintf(stderr, "lpp: t-J hole-major form: %d hole configurations x %d spin patterns in %zu items (segments of the low %d positions), %zu bonds, %zu moves, %.2f MB of tables; against the row walk: largest difference %.3g of %.3g\n",
		        nblk, ns, items.size(), kbits, pairs.size(), hops.size(), 1e-6 * (double)S.table_bytes, dmax, xmax);
	if (!(dmax <= 1e-12 * std::max(xmax, 1e-300))) return LPP_OK; // not the same matrix: the general layout (the guard drops this one)
	S.nnz = (int64_t)nnz;
	S.active = true;
	e->pitch = S.pitch;
	e->pitch_rows = ns;
	e->pitch_blocks = nblk;
	*done = true;
	return LPP_OK;
}

} // namespace lpp
