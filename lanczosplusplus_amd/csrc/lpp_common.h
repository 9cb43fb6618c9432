// lpp_common.h -- constants, wave/block reductions, value traits and the SpMV epilogue scaling shared by all kernels.
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
	static __device__ __forceinline__ double sub_scaled(double x, double s, double y) { return x - s * y; }
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
	static __device__ __forceinline__ cplx sub_scaled(cplx x, double s, cplx y) { return cplx { x.re - s * y.re, x.im - s * y.im }; }
	static __device__ __forceinline__ cplx shfl_down(cplx v, int off, int w)
	{
		return cplx { __shfl_down(v.re, off, w), __shfl_down(v.im, off, w) };
	}
};

// Epilogue scaling of the SpMV kernels:  x_new = beta * x_old + alpha * (H y)_row.
// Plain products use alpha = beta = 1 (x += H y).  The scale-free Lanczos recurrence keeps the Lanczos
// vectors unnormalised (r_j = b_{j-1} y_j) and folds the scalings into this epilogue:
// alpha = 1/b_{j-1}, beta = -b_{j-1}/b_{j-2}, both derived in-kernel from b^2 values in device memory,
// which removes the separate swap/scale pass (4 N s bytes per step).
struct EpiScale {
	const double* b2_prev; // b_{j-1}^2 (null: alpha = 1)
	const double* b2_prev2; // b_{j-2}^2 (null: beta = 0 when b2_prev is set)
	int beta_one; // 1: beta = 1 regardless (second kernel of a split product)
};

__device__ __forceinline__ void epi_coeffs(const EpiScale& sc, double& alpha, double& beta)
{
	alpha = 1.0;
	beta = 1.0;
	if (sc.b2_prev) {
		const double b1 = sqrt(*sc.b2_prev);
		alpha = (fabs(b1) < 1e-10) ? 1.0 : 1.0 / b1;
		if (!sc.beta_one) {
			beta = 0.0;
			if (sc.b2_prev2) {
				const double b2 = sqrt(*sc.b2_prev2);
				beta = (fabs(b2) < 1e-10) ? -b1 : -b1 / b2;
			}
		}
	}
}

__device__ __forceinline__ double epi_lin(double beta, double xold, double alpha, double acc) { return beta * xold + alpha * acc; }
__device__ __forceinline__ cplx epi_lin(double beta, cplx xold, double alpha, cplx acc)
{
	return cplx { beta * xold.re + alpha * acc.re, beta * xold.im + alpha * acc.im };
}

} // namespace lpp
