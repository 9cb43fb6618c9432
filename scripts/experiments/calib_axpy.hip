// calib_axpy.hip -- ceiling of the 2-read 1-write pass x -= a y (+ |x|^2) of k_axpy_nrm at two vector lengths: config 2 (1.33 GB
// per vector, beyond the 256 MiB Infinity Cache) and config 3 (Heisenberg L = 28: 0.32 GB).   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k(d2* __restrict__ x, const d2* __restrict__ y, long n2, double a, double* part)
{
	double s = 0;
	const long stride = (long)gridDim.x * 256;
	long i = (long)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n2; i += U * stride) {
		d2 xv[U], yv[U];
#pragma unroll
		for (int q = 0; q < U; q++) xv[q] = NT ? __builtin_nontemporal_load(&x[i + q * stride]) : x[i + q * stride];
#pragma unroll
		for (int q = 0; q < U; q++) yv[q] = NT ? __builtin_nontemporal_load(&y[i + q * stride]) : y[i + q * stride];
#pragma unroll
		for (int q = 0; q < U; q++) {
			d2 r = xv[q] - a * yv[q];
			if (NT) __builtin_nontemporal_store(r, &x[i + q * stride]); else x[i + q * stride] = r;
			s += r.x * r.x + r.y * r.y;
		}
	}
	for (; i < n2; i += stride) { d2 r = x[i] - a * y[i]; x[i] = r; s += r.x * r.x + r.y * r.y; }
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x & 1023], s);
}
// contiguous chunk per block instead of a grid stride
template <int U, bool NT> __global__ __launch_bounds__(256) void kc(d2* __restrict__ x, const d2* __restrict__ y, long n2, double a, double* part)
{
	double s = 0;
	const long per = (n2 + gridDim.x - 1) / gridDim.x;
	const long lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
	long i = lo + threadIdx.x;
	for (; i + (U - 1) * 256 < hi; i += U * 256) {
		d2 xv[U], yv[U];
#pragma unroll
		for (int q = 0; q < U; q++) xv[q] = NT ? __builtin_nontemporal_load(&x[i + q * 256]) : x[i + q * 256];
#pragma unroll
		for (int q = 0; q < U; q++) yv[q] = NT ? __builtin_nontemporal_load(&y[i + q * 256]) : y[i + q * 256];
#pragma unroll
		for (int q = 0; q < U; q++) {
			d2 r = xv[q] - a * yv[q];
			if (NT) __builtin_nontemporal_store(r, &x[i + q * 256]); else x[i + q * 256] = r;
			s += r.x * r.x + r.y * r.y;
		}
	}
	for (; i < hi; i += 256) { d2 r = x[i] - a * y[i]; x[i] = r; s += r.x * r.x + r.y * r.y; }
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x & 1023], s);
}
template <typename F> void run(const char* name, int U, int nt, F launch, long n2, int blocks)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int w = 0; w < 2; w++) launch(blocks);
	hipEventRecord(a);
	const int it = 10;
	for (int w = 0; w < it; w++) launch(blocks);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
	printf("%-8s U=%d nt=%d blocks=%5d  %.3f ms  %.2f TB/s\n", name, U, nt, blocks, ms, 3.0 * n2 * 16 / ms / 1e9);
}
int main()
{
	for (long n2 : { 12870L * 12880 / 2, 40116600L / 2 }) {
		printf("n2 = %ld (%.2f GB per vector)\n", n2, n2 * 16 / 1e9);
		d2 *x, *y; double* part;
		hipMalloc(&x, n2 * 16); hipMalloc(&y, n2 * 16); hipMalloc(&part, 8192);
		hipMemset(x, 0, n2 * 16); hipMemset(y, 0, n2 * 16); hipMemset(part, 0, 8192);
		for (int blocks : { 2048, 4096, 8192, 16384 }) {
#define RUN(K, NAME, U_, NT_) run(NAME, U_, NT_, [&](int nb) { K<U_, NT_><<<nb, 256>>>(x, y, n2, 1e-3, part); }, n2, blocks)
			RUN(k, "stride", 2, true);
			RUN(k, "stride", 4, true);
			RUN(k, "stride", 8, true);
			RUN(k, "stride", 4, false);
			RUN(kc, "chunk", 4, true);
			RUN(kc, "chunk", 4, false);
		}
		hipFree(x); hipFree(y); hipFree(part);
	}
	return 0;
}
