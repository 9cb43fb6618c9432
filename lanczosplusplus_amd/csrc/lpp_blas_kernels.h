// lpp_blas_kernels.h -- fused BLAS-1 of the three-term recurrence, the transposition exchange pack/unpack, blocked
// Gram-Schmidt panels, the deterministic second reduction stage and the start-vector generator.
#pragma once
#include "lpp_common.h"

namespace lpp {

// ---------------------------------------------------------------------------------------------
// fused BLAS-1 of the three-term recurrence (double2 = 16 B per lane)
// ---------------------------------------------------------------------------------------------

// x -= g*y ;  partial[b] = sum |x|^2     (scalars read from device memory: no host round trip)
// g = *a_ptr (normalised recurrence) or *a_ptr / *b2_prev (scale-free recurrence: raw dot <r_j|w> over b_{j-1}^2).
// `send` (optional) receives a copy of the new x: the slice handed to the next all-gather.
// streamed 16-byte accesses (read once / written once per pass: keep them out of the way of the SpMV's L2 contents)
typedef double lpp_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2* p)
{
	const lpp_d2 v = __builtin_nontemporal_load((const lpp_d2*)p);
	return double2 { v.x, v.y };
}
__device__ __forceinline__ void nt_store2(double2 v, double2* p)
{
	lpp_d2 w;
	w.x = v.x;
	w.y = v.y;
	__builtin_nontemporal_store(w, (lpp_d2*)p);
}

// copy for the next exchange: the send buffer holds exactly nd doubles, so an odd tail element is stored as a scalar
__device__ __forceinline__ void send_store(double2* send, int64_t i, double2 v, int64_t nd)
{
	if (2 * i + 2 <= nd)
		send[i] = v;
	else if (2 * i < nd)
		((double*)send)[2 * i] = v.x;
}

template <bool NRM>
__global__ __launch_bounds__(kBlock) void k_axpy_nrm(double2* __restrict__ x, const double2* __restrict__ y,
                                                      const double* __restrict__ a_ptr, const double* __restrict__ b2_prev,
                                                      double2* __restrict__ send, int64_t n2, double* __restrict__ partial, int stream = 0, int64_t nd = 0,
                                                      const double2* __restrict__ zadd = nullptr)
{
	__shared__ double smem[kBlock / 64];
	double a = *a_ptr;
	if (b2_prev) {
		const double b2 = *b2_prev;
		if (sqrt(b2) >= 1e-10) a /= b2;
	}
	double s = 0.0;
	const int64_t stride = (int64_t)gridDim.x * kBlock;
	int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
	// four independent 16-byte loads of x and of y in flight per lane (a one-element loop body left the kernel at
	// 4.5 TB/s: the loads of the next iteration were not issued before the store of this one)
	for (; i + 3 * stride < n2; i += 4 * stride) {
		double2 xv[4], yv[4];
#pragma unroll
		for (int k = 0; k < 4; k++) xv[k] = stream ? nt_load2(&x[i + k * stride]) : x[i + k * stride];
#pragma unroll
		for (int k = 0; k < 4; k++) yv[k] = stream ? nt_load2(&y[i + k * stride]) : y[i + k * stride];
		if (zadd) { // a part of the product that was formed in a buffer of its own (product-basis layout: the block couplings)
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const double2 zv = nt_load2(&zadd[i + k * stride]);
				xv[k].x += zv.x;
				xv[k].y += zv.y;
			}
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			xv[k].x -= a * yv[k].x;
			xv[k].y -= a * yv[k].y;
			if (stream) // vectors beyond the Infinity Cache: nothing of this pass is re-read before it is evicted anyway
				nt_store2(xv[k], &x[i + k * stride]);
			else
				x[i + k * stride] = xv[k];
			if (send) send_store(send, i + k * stride, xv[k], nd);
			if (NRM) s += xv[k].x * xv[k].x + xv[k].y * xv[k].y;
		}
	}
	for (; i < n2; i += stride) {
		double2 xv = x[i];
		const double2 yv = y[i];
		if (zadd) {
			xv.x += zadd[i].x;
			xv.y += zadd[i].y;
		}
		xv.x -= a * yv.x;
		xv.y -= a * yv.y;
		x[i] = xv;
		if (send) send_store(send, i, xv, nd);
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
                                                        const double* __restrict__ b2_ptr, int64_t n2, int64_t nd = 0)
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
		if (send) send_store(send, i, yn, nd);
	}
}

// dst = src / sqrt(*n2_ptr)   (normalise the start vector); optional second copy
static __global__ __launch_bounds__(kBlock) void k_scale_copy(double2* __restrict__ dst, double2* __restrict__ send,
                                                        const double2* __restrict__ src,
                                                        const double* __restrict__ nrm2_ptr, int64_t n2, int64_t nd = 0)
{
	const double inv = 1.0 / sqrt(*nrm2_ptr);
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
		double2 v = src[i];
		v.x *= inv;
		v.y *= inv;
		dst[i] = v;
		if (send) send_store(send, i, v, nd);
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

// ---------------------------------------------------------------------------------------------
// Transposition exchange (multi-GPU Hubbard): the rank's slice y[(id-id0)*N_up + iu] is re-cut by UP index.
// Chunk p of the send buffer holds the sub-block iu in [p*peru, (p+1)*peru) of every local down index:
//   send[p*C + id_l*peru + iu_lp],  C = per*peru (padded, padding never written and pre-zeroed).
// After the all-to-all, chunk q of the receive buffer holds the rank's own UP range for rank q's down indices,
// i.e. the transposed slice yT[id*peru + iu_l] with id running over ALL down indices.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_transpose(const T* __restrict__ y, T* __restrict__ send, int64_t nid,
                                                            int64_t n_up, int64_t peru, int64_t chunk, int64_t pitch = 0)
{
	const int64_t n = nid * n_up;
	if (pitch == 0) pitch = n_up; // pitch > n_up: the slice is stored with padded rows (product-basis layout)
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
		const int64_t idl = i / n_up, iu = i - idl * n_up;
		const int64_t p = iu / peru, iul = iu - p * peru;
		send[p * chunk + idl * peru + iul] = y[idl * pitch + iu];
	}
}

// The same with the deferred recurrence update folded in (product-basis layout, scale-free recurrence): the slice still holds
// w = H r/b - (b/b') r'; r_next = w - g r (g = *g_a / *g_b2, as k_axpy_nrm has it) is formed on the way, written back over w and
// packed -- one pass and one launch less than k_axpy_nrm followed by k_pack_transpose.  Padding elements are not touched.
static __global__ __launch_bounds__(kBlock) void k_pack_transpose_axpy(double* __restrict__ w, const double* __restrict__ r, const double* __restrict__ g_a,
                                                                        const double* __restrict__ g_b2, double* __restrict__ send, int64_t nid, int64_t n_up,
                                                                        int64_t peru, int64_t chunk, int64_t pitch)
{
	double g = *g_a;
	const double b2 = *g_b2;
	if (sqrt(b2) >= 1e-10) g /= b2;
	const int64_t n = nid * n_up;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
		const int64_t idl = i / n_up, iu = i - idl * n_up;
		const int64_t p = iu / peru, iul = iu - p * peru;
		const int64_t k = idl * pitch + iu;
		const double v = w[k] - g * r[k];
		w[k] = v;
		send[p * chunk + idl * peru + iul] = v;
	}
}

// x[i] += recv[...] (the down-hop part computed on the UP-partitioned layout and sent back), fused Re<y|x> partial
template <typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void k_unpack_add_dot(T* __restrict__ x, const T* __restrict__ recv,
                                                            const T* __restrict__ y, int64_t nid, int64_t n_up,
                                                            int64_t peru, int64_t chunk, double* __restrict__ partial,
                                                            const double* __restrict__ shift = nullptr)
{
	__shared__ double smem[kBlock / 64];
	const int64_t n = nid * n_up;
	const double sh = shift ? *shift : 0.0;
	double dot = 0.0, nrm = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
		const int64_t idl = i / n_up, iu = i - idl * n_up;
		const int64_t p = iu / peru, iul = iu - p * peru;
		const T xv = VT<T>::add(x[i], recv[p * chunk + idl * peru + iul]);
		x[i] = xv;
		if (DOT) {
			const T yv = y[i];
			const T d = VT<T>::sub_scaled(xv, sh, yv);
			dot += VT<T>::dot_re(yv, xv);
			nrm += VT<T>::dot_re(d, d);
		}
	}
	if (DOT) { // partial[2b] = Re<y|x>, partial[2b+1] = |x - s y|^2: both travel in ONE all-reduce (see k_b2_from_w)
		const double r = block_sum(dot, smem);
		if (threadIdx.x == 0) partial[2 * blockIdx.x] = r;
		const double q = block_sum(nrm, smem);
		if (threadIdx.x == 0) partial[2 * blockIdx.x + 1] = q;
	}
}

// One reduction per Lanczos step: b_j^2 without a pass (and an all-reduce) of its own.
// In the scale-free recurrence r_{j+1} = w - (raw/b_{j-1}^2) r_j with raw = Re<r_j|w> and |r_j|^2 = b_{j-1}^2, and for ANY real s
//     b_j^2 = |r_{j+1}|^2 = |w - s r_j|^2 - (raw - s b_{j-1}^2)^2 / b_{j-1}^2
// (r_{j+1} does not depend on a shift of H).  With s = 0 this is |w|^2 - a_j^2: it cancels, and worse, the error of the previous
// norm enters multiplied by a_j^2 / b_j^2 -- measured on the 12-site Hubbard chain (a = 12, b = 9): the coefficients drift apart
// by a factor 1.7 per step and the run stops 16 steps early, 1.4e-7 off.  With s = a_{j-1} / b_{j-1} (last step's diagonal
// coefficient as the shift) the subtracted term is (a_j - a_{j-1})^2, far below b_j^2, and the recursion contracts.
// The product's last kernel therefore sums Re<r_j|w> and |w - s r_j|^2 ELEMENTWISE (s from *shift); both are reduced (over the
// ranks: in one all-reduce) and this kernel turns ab[1] into b_j^2 and leaves the next step's s in *shift.
static __global__ void k_b2_from_w(double* __restrict__ ab, const double* __restrict__ b2_prev, double* __restrict__ shift)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		const double raw = ab[0], ws = ab[1], bp = *b2_prev, s = *shift;
		double b2 = ws;
		const bool ok = sqrt(bp) >= 1e-10;
		if (ok) {
			const double rs = raw - s * bp;
			b2 = ws - rs * rs / bp;
		}
		b2 = b2 > 0.0 ? b2 : 0.0;
		ab[1] = b2;
		*shift = (ok && sqrt(b2) >= 1e-10) ? raw / (sqrt(bp) * sqrt(b2)) : 0.0;
	}
}

// splitmix64 start vector; the test-suite checks it is bit-identical to the CPU checker's stream
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
