// calib_combine.hip -- how fast can the streaming pass x = beta x + u + z - g y (4 reads + 1 write, 16 B per lane) go?
// Variants: elements in flight per lane and stream (U), grid size, non-temporal vs plain accesses.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k(d2* __restrict__ x, const d2* __restrict__ y, const d2* __restrict__ u, const d2* __restrict__ z, long n2, double beta, double g, double* part)
{
	double s = 0;
	const long stride = (long)gridDim.x * 256;
	long i = (long)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n2; i += U * stride) {
		d2 xv[U], yv[U], uv[U], zv[U];
#pragma unroll
		for (int k = 0; k < U; k++) xv[k] = NT ? __builtin_nontemporal_load(&x[i + k * stride]) : x[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) yv[k] = NT ? __builtin_nontemporal_load(&y[i + k * stride]) : y[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) uv[k] = NT ? __builtin_nontemporal_load(&u[i + k * stride]) : u[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) zv[k] = NT ? __builtin_nontemporal_load(&z[i + k * stride]) : z[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) {
			d2 r = beta * xv[k] + uv[k] + zv[k] - g * yv[k];
			if (NT) __builtin_nontemporal_store(r, &x[i + k * stride]); else x[i + k * stride] = r;
			s += r.x * r.x + r.y * r.y;
		}
	}
	for (; i < n2; i += stride) { d2 r = beta * x[i] + u[i] + z[i] - g * y[i]; x[i] = r; s += r.x * r.x + r.y * r.y; }
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(&part[blockIdx.x & 1023], s);
}
template <int U, bool NT> void run(const char* name, d2* x, d2* y, d2* u, d2* z, long n2, int blocks, double* part)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int w = 0; w < 2; w++) k<U, NT><<<blocks, 256>>>(x, y, u, z, n2, 0.5, 0.1, part);
	hipEventRecord(a);
	const int it = 10;
	for (int w = 0; w < it; w++) k<U, NT><<<blocks, 256>>>(x, y, u, z, n2, 0.5, 0.1, part);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
	printf("%-10s U=%d nt=%d blocks=%5d  %.3f ms  %.2f TB/s\n", name, U, (int)NT, blocks, ms, 5.0 * n2 * 16 / ms / 1e9);
}
int main()
{
	const long n2 = 12870L * 12880 / 2; // config 2's pitched vector in 16-byte units
	d2 *x, *y, *u, *z; double* part;
	hipMalloc(&x, n2 * 16); hipMalloc(&y, n2 * 16); hipMalloc(&u, n2 * 16); hipMalloc(&z, n2 * 16); hipMalloc(&part, 8192);
	hipMemset(x, 0, n2 * 16); hipMemset(y, 0, n2 * 16); hipMemset(u, 0, n2 * 16); hipMemset(z, 0, n2 * 16); hipMemset(part, 0, 8192);
	for (int blocks : { 1024, 2048, 4096, 8192, 16384 }) {
		run<1, true>("combine", x, y, u, z, n2, blocks, part);
		run<2, true>("combine", x, y, u, z, n2, blocks, part);
		run<4, true>("combine", x, y, u, z, n2, blocks, part);
		run<2, false>("combine", x, y, u, z, n2, blocks, part);
		run<4, false>("combine", x, y, u, z, n2, blocks, part);
	}
	return 0;
}
