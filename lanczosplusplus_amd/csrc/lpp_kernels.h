// lpp_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the Lanczos inner engine.
//
// Every kernel here is HBM-bandwidth bound (SpMV: ~0.17 flop/byte); there is deliberately no MFMA.
// Conventions: vectors are arrays of doubles padded to an even count so BLAS-1 kernels move
// 16 B per lane (double2); complex values are interleaved (re,im) == one double2.
// Reductions are two-stage (per-block partial -> k_reduce_final) so results are bitwise
// reproducible run to run (no float atomics).
//
// Reference semantics restated (all under /root/reference/src):
//   x += H y                         Engine/InternalProductStored.h:77,121-124
//   a = Re<y|x>; x -= a y; b = |x|;  (y,x) <- (x/b, -b y)     LanczosSolver [PsimagLite], SURVEY 3.1
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lpp {

struct __attribute__((aligned(16))) cplx {
	double re, im;
};

constexpr int kBlock = 256; // 4 waves
constexpr int kMaxPartials = 4096; // upper bound on blocks of any reducing kernel
constexpr int kPanel = 8; // Gram-Schmidt panel width

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave64: hard-coded 64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

// result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* smem)
{
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	__syncthreads(); // protect smem reuse across consecutive calls
	if (lane == 0) smem[w] = v;
	__syncthreads();
	double r = 0;
	if (threadIdx.x == 0) {
#pragma unroll
		for (int i = 0; i < kBlock / 64; i++) r += smem[i];
	}
	return r;
}

// ---------------------------------------------------------------------------------------------
// value traits
// ---------------------------------------------------------------------------------------------
template <typename T> struct VT;
template <> struct VT<double> {
	static __device__ __forceinline__ double zero() { return 0.0; }
	static __device__ __forceinline__ void mac(double& acc, double v, double y) { acc += v * y; }
	static __device__ __forceinline__ double add(double a, double b) { return a + b; }
	static __device__ __forceinline__ double dot_re(double y, double x) { return y * x; } // Re(y conj x)
	static __device__ __forceinline__ double shfl_down(double v, int off, int w) { return __shfl_down(v, off, w); }
};
template <> struct VT<cplx> {
	static __device__ __forceinline__ cplx zero() { return cplx { 0.0, 0.0 }; }
	static __device__ __forceinline__ void mac(cplx& acc, cplx v, cplx y)
	{
		acc.re += v.re * y.re - v.im * y.im;
		acc.im += v.re * y.im + v.im * y.re;
	}
	static __device__ __forceinline__ cplx add(cplx a, cplx b) { return cplx { a.re + b.re, a.im + b.im }; }
	static __device__ __forceinline__ double dot_re(cplx y, cplx x) { return y.re * x.re + y.im * x.im; }
	static __device__ __forceinline__ cplx shfl_down(cplx v, int off, int w)
	{
		return cplx { __shfl_down(v.re, off, w), __shfl_down(v.im, off, w) };
	}
};

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
                                                           double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	const int lig = threadIdx.x % G;
	const int64_t ngroups = (int64_t)gridDim.x * (kBlock / G);
	double dot = 0.0;
	for (int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G; row < nrows; row += ngroups) {
		const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
		T acc = VT<T>::zero();
		for (int64_t p = p0 + lig; p < p1; p += 4 * G) {
			int32_t c[4];
			T v[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int64_t pk = p + (int64_t)k * G;
				const bool ok = pk < p1;
				c[k] = ok ? col[pk] : -1;
				v[k] = ok ? val[pk] : VT<T>::zero();
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (c[k] >= 0) VT<T>::mac(acc, v[k], src[c[k]]);
		}
#pragma unroll
		for (int off = G / 2; off > 0; off >>= 1) acc = VT<T>::add(acc, VT<T>::shfl_down(acc, off, G));
		if (lig == 0) {
			const T xv = VT<T>::add(x[row], acc);
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
// K2: sliced ("wave-interleaved") CSR SpMV.
//
// Device-internal layout built once from the CSR (k_slice_*): rows are grouped in slices of 64
// (one wave); inside a slice the entries are stored slot-major and COMPACT: all first entries of
// the rows that have one, then all second entries, ... -- same bytes as CSR, no padding, no row
// permutation.  Lane r owns row slice*64+r; at slot k the active lanes (len > k) read a dense,
// coalesced run of val/col, the position of a lane inside the run being the popcount of the
// active mask below it (ballot + mbcnt).  The gather src[col] then has lanes = consecutive rows,
// which for product bases (Hubbard down-hops: col = row + const*N_up) is itself coalesced.
// ---------------------------------------------------------------------------------------------
template <typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_sliced(int64_t nrows, int64_t nslices,
                                                         const int64_t* __restrict__ slice_ptr,
                                                         const int32_t* __restrict__ row_len,
                                                         const int32_t* __restrict__ col,
                                                         const T* __restrict__ val, const T* __restrict__ src,
                                                         T* __restrict__ x, const T* __restrict__ ydot,
                                                         double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	double dot = 0.0;
	for (int64_t s = wave0; s < nslices; s += nwaves) {
		const int64_t row = s * 64 + lane;
		const int len = (row < nrows) ? row_len[row] : 0;
		int64_t base = slice_ptr[s];
		T acc = VT<T>::zero();
		// max length in the wave
		int maxlen = len;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
		int k = 0;
		for (; k + 4 <= maxlen; k += 4) { // 4 independent slots in flight
			int32_t c[4];
			T v[4];
			bool on[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				on[u] = len > k + u;
				const unsigned long long m = __ballot(on[u]);
				const int pos = __popcll(m & ((1ull << lane) - 1ull));
				const int64_t p = base + pos;
				c[u] = on[u] ? col[p] : 0;
				v[u] = on[u] ? val[p] : VT<T>::zero();
				base += __popcll(m);
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (on[u]) VT<T>::mac(acc, v[u], src[c[u]]);
		}
		for (; k < maxlen; k++) {
			const bool on = len > k;
			const unsigned long long m = __ballot(on);
			const int pos = __popcll(m & ((1ull << lane) - 1ull));
			if (on) {
				const int64_t p = base + pos;
				VT<T>::mac(acc, val[p], src[col[p]]);
			}
			base += __popcll(m);
		}
		if (row < nrows) {
			const T xv = VT<T>::add(x[row], acc);
			x[row] = xv;
			if (DOT) dot += VT<T>::dot_re(ydot[row], xv);
		}
	}
	if (DOT) {
		const double r = block_sum(dot, smem);
		if (threadIdx.x == 0) partial[blockIdx.x] = r;
	}
}

// CSR -> sliced layout: per-slice sizes are rowptr differences (host/thrust-free: computed by
// k_slice_ptr from rowptr directly since slices are contiguous row ranges).
static __global__ void k_slice_meta(int64_t nrows, int64_t nslices, const int64_t* __restrict__ rowptr,
                             int64_t* __restrict__ slice_ptr, int32_t* __restrict__ row_len)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nrows) row_len[i] = (int32_t)(rowptr[i + 1] - rowptr[i]);
	if (i <= nslices) {
		const int64_t r = (i * 64 < nrows) ? i * 64 : nrows;
		slice_ptr[i] = rowptr[r];
	}
}

// one wave per slice: scatter CSR entries into slot-major compact order
template <typename T>
__global__ __launch_bounds__(kBlock) void k_slice_fill(int64_t nrows, int64_t nslices,
                                                        const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col_in,
                                                        const T* __restrict__ val_in, int32_t* __restrict__ col_out,
                                                        T* __restrict__ val_out)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
	for (int64_t s = wave0; s < nslices; s += nwaves) {
		const int64_t row = s * 64 + lane;
		int64_t p0 = 0;
		int len = 0;
		if (row < nrows) {
			p0 = rowptr[row];
			len = (int)(rowptr[row + 1] - p0);
		}
		int64_t base = rowptr[s * 64];
		int maxlen = len;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
		for (int k = 0; k < maxlen; k++) {
			const bool on = len > k;
			const unsigned long long m = __ballot(on);
			const int pos = __popcll(m & ((1ull << lane) - 1ull));
			if (on) {
				col_out[base + pos] = col_in[p0 + k];
				val_out[base + pos] = val_in[p0 + k];
			}
			base += __popcll(m);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// fused BLAS-1 of the three-term recurrence (double2 = 16 B per lane)
// ---------------------------------------------------------------------------------------------

// x -= a*y ;  partial[b] = sum |x|^2     (a read from device memory: no host round trip)
template <bool NRM>
__global__ __launch_bounds__(kBlock) void k_axpy_nrm(double2* __restrict__ x, const double2* __restrict__ y,
                                                      const double* __restrict__ a_ptr, int64_t n2,
                                                      double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	const double a = *a_ptr;
	double s = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		double2 xv = x[i];
		const double2 yv = y[i];
		xv.x -= a * yv.x;
		xv.y -= a * yv.y;
		x[i] = xv;
		if (NRM) s += xv.x * xv.x + xv.y * xv.y;
	}
	if (NRM) {
		const double r = block_sum(s, smem);
		if (threadIdx.x == 0) partial[blockIdx.x] = r;
	}
}

static __global__ __launch_bounds__(kBlock) void k_dot(const double2* __restrict__ x, const double2* __restrict__ y,
                                                 int64_t n2, double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	double s = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const double2 xv = x[i], yv = y[i];
		s += xv.x * yv.x + xv.y * yv.y;
	}
	const double r = block_sum(s, smem);
	if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// (y_next, x) <- (x / b, -b * y)  with b = sqrt(*b2_ptr);  |b| < 1e-10 leaves x unscaled
// (the reference's guard in LanczosSolver::oneStepDecomposition [PsimagLite]).
// `send` (optional) receives a second copy of y_next: the slice handed to the all-gather.
// y and ynext may alias (in-place swap when the Lanczos vectors are not kept), hence no __restrict__.
static __global__ __launch_bounds__(kBlock) void k_swap_scale(double2* __restrict__ x, const double2* y,
                                                        double2* ynext, double2* __restrict__ send,
                                                        const double* __restrict__ b2_ptr, int64_t n2)
{
	const double b = sqrt(*b2_ptr);
	const double inv = (fabs(b) < 1e-10) ? 1.0 : 1.0 / b;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const double2 xv = x[i];
		const double2 yv = y[i];
		double2 yn, xn;
		yn.x = xv.x * inv;
		yn.y = xv.y * inv;
		xn.x = -b * yv.x;
		xn.y = -b * yv.y;
		ynext[i] = yn;
		x[i] = xn;
		if (send) send[i] = yn;
	}
}

// dst = src / sqrt(*n2_ptr)   (normalise the start vector); optional second copy
static __global__ __launch_bounds__(kBlock) void k_scale_copy(double2* __restrict__ dst, double2* __restrict__ send,
                                                        const double2* __restrict__ src,
                                                        const double* __restrict__ nrm2_ptr, int64_t n2)
{
	const double inv = 1.0 / sqrt(*nrm2_ptr);
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		double2 v = src[i];
		v.x *= inv;
		v.y *= inv;
		dst[i] = v;
		if (send) send[i] = v;
	}
}

// z += s * y   (two-pass Ritz accumulation; s passed by value)
static __global__ __launch_bounds__(kBlock) void k_axpy_const(double2* __restrict__ z, const double2* __restrict__ y,
                                                        double s, int64_t n2)
{
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		double2 zv = z[i];
		const double2 yv = y[i];
		zv.x += s * yv.x;
		zv.y += s * yv.y;
		z[i] = zv;
	}
}

// splitmix64 start vector, bit-identical to oracle/lpp_oracle.c:lppo_fill_random
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
static __global__ void k_fill_random(double* __restrict__ v, int64_t nd, int64_t offset, uint64_t seed)
{
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t r = splitmix64(seed * 0x2545F4914F6CDD1DULL + (uint64_t)(k + offset));
		v[k] = (double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5;
	}
}

// out[c] = sum_p partial[p*stride + c], c < count  (single block; fixed summation order)
static __global__ __launch_bounds__(kBlock) void k_reduce_final(const double* __restrict__ partial, int np, int stride,
                                                          int count, double* __restrict__ out)
{
	__shared__ double smem[kBlock / 64];
	for (int c = 0; c < count; c++) {
		double s = 0.0;
		for (int p = threadIdx.x; p < np; p += kBlock) s += partial[(int64_t)p * stride + c];
		const double r = block_sum(s, smem);
		if (threadIdx.x == 0) out[c] = r;
	}
}

// ---------------------------------------------------------------------------------------------
// blocked Gram-Schmidt against the on-device Krylov basis (panels of kPanel columns)
//   coef_p = <v_p | x> = sum conj(v_p) x        k_multi_dot   (reads x once per panel)
//   x     -= sum_p coef_p v_p                   k_multi_axpy
// V column p of the panel starts at v0 + p*ldv (ldv in double2 units).
// ---------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_multi_dot(const double2* __restrict__ x, const double2* __restrict__ v0,
                                                       int64_t ldv, int np, int64_t n2,
                                                       double* __restrict__ partial)
{
	__shared__ double smem[kBlock / 64];
	double re[kPanel], im[kPanel];
#pragma unroll
	for (int p = 0; p < kPanel; p++) re[p] = im[p] = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		const double2 xv = x[i];
#pragma unroll
		for (int p = 0; p < kPanel; p++) {
			if (p < np) {
				const double2 vv = v0[(int64_t)p * ldv + i];
				re[p] += vv.x * xv.x + vv.y * xv.y;
				if (CPLX) im[p] += vv.x * xv.y - vv.y * xv.x;
			}
		}
	}
#pragma unroll
	for (int p = 0; p < kPanel; p++) {
		const double r = block_sum(re[p], smem);
		if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * (2 * kPanel) + 2 * p] = r;
		const double q = CPLX ? block_sum(im[p], smem) : 0.0;
		if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * (2 * kPanel) + 2 * p + 1] = q;
	}
}

// coef: 2 doubles (re,im) per panel column in device memory; sign = -1 for orthogonalisation,
// +1 to accumulate Ritz vectors (z += sum S_jk v_j).
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_multi_axpy(double2* __restrict__ x, const double2* __restrict__ v0,
                                                        int64_t ldv, int np, const double* __restrict__ coef,
                                                        double sign, int64_t n2)
{
	double cr[kPanel], ci[kPanel];
#pragma unroll
	for (int p = 0; p < kPanel; p++) {
		cr[p] = (p < np) ? sign * coef[2 * p] : 0.0;
		ci[p] = (p < np && CPLX) ? sign * coef[2 * p + 1] : 0.0;
	}
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		double2 xv = x[i];
#pragma unroll
		for (int p = 0; p < kPanel; p++) {
			if (p < np) {
				const double2 vv = v0[(int64_t)p * ldv + i];
				if (CPLX) {
					xv.x += cr[p] * vv.x - ci[p] * vv.y;
					xv.y += cr[p] * vv.y + ci[p] * vv.x;
				} else {
					xv.x += cr[p] * vv.x;
					xv.y += cr[p] * vv.y;
				}
			}
		}
		x[i] = xv;
	}
}

} // namespace lpp
