// What a CU's vector-memory pipeline charges per wave-level load: width (4 / 8 / 16 bytes per lane) x pattern (64 consecutive elements /
// 64 different 128-byte lines) x footprint (L1-resident 16 KB per CU, L2-resident 2 MB, 512 MB).  One 1024-thread workgroup per CU,
// 8 independent loads in flight per wave and iteration.  Prints cycles per wave-level instruction and CU, and GB/s per CU.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/tcp_rate scripts/experiments/r04_tcp_rate.hip && /tmp/tcp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <typename T> __device__ inline double as_sum(const T& v);
template <> __device__ inline double as_sum<uint32_t>(const uint32_t& v) { return (double)v; }
template <> __device__ inline double as_sum<double>(const double& v) { return v; }
template <> __device__ inline double as_sum<double2>(const double2& v) { return v.x + v.y; }

// lanes read element (lane * lane_stride + k * step + it * 8 * step) mod n  of the workgroup's region
template <typename T> __global__ __launch_bounds__(1024) void k_rate(const T* buf, size_t n_per_wg, int lane_stride, int step, int iters, double* out, int shared_region)
{
	const T* base = buf + (shared_region ? 0 : (size_t)blockIdx.x * n_per_wg);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	size_t idx = ((size_t)lane * lane_stride + (size_t)wave * 64 * lane_stride) % n_per_wg;
	double acc = 0.0;
	for (int it = 0; it < iters; it++) {
		T v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			v[k] = base[idx];
			idx += step;
			if (idx >= n_per_wg) idx -= n_per_wg;
		}
#pragma unroll
		for (int k = 0; k < 8; k++) acc += as_sum<T>(v[k]);
	}
	if (acc == 1.2345e-300) out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <typename T> static void run(const char* name, const void* buf, size_t bytes_per_wg, int lane_stride_elems, int step_elems, int shared, int ncu, double mhz, double* out)
{
	const size_t n = bytes_per_wg / sizeof(T);
	const int iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	k_rate<T><<<ncu, 1024>>>((const T*)buf, n, lane_stride_elems, step_elems, 10, out, shared);
	hipEventRecord(a);
	k_rate<T><<<ncu, 1024>>>((const T*)buf, n, lane_stride_elems, step_elems, iters, out, shared);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double instr_per_cu = 16.0 * iters * 8;
	const double cyc = ms * 1e-3 * mhz * 1e6;
	printf("%-58s %8.1f cycles/instr/CU  %7.1f GB/s/CU\n", name, cyc / instr_per_cu, instr_per_cu * 64 * sizeof(T) / (ms * 1e-3) / 1e9);
}

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int ncu = p.multiProcessorCount;
	const double mhz = p.clockRate / 1e3;
	printf("%s: %d CUs, %.0f MHz\n", p.name, ncu, mhz);
	const size_t big = (size_t)512 << 20;
	void* buf;
	hipMalloc(&buf, big);
	hipMemset(buf, 0, big);
	double* out;
	hipMalloc(&out, sizeof(double) * 1024 * ncu);
	struct { const char* fp; size_t per_wg; int shared; } fps[] = { { "16 KB per CU (L1)", 16 << 10, 0 }, { "2 MB shared (L2)", 2 << 20, 1 }, { "512 MB shared (HBM / Infinity Cache)", big, 1 } };
	for (auto& f : fps) {
		char nm[128];
		// consecutive lanes, consecutive elements; the next load of a wave 16 waves x 64 elements further on
		snprintf(nm, sizeof nm, "%s, 4 B/lane coalesced (256 B/instr)", f.fp);
		run<uint32_t>(nm, buf, f.per_wg, 1, 1024, f.shared, ncu, mhz, out);
		snprintf(nm, sizeof nm, "%s, 8 B/lane coalesced (512 B/instr)", f.fp);
		run<double>(nm, buf, f.per_wg, 1, 1024, f.shared, ncu, mhz, out);
		snprintf(nm, sizeof nm, "%s, 16 B/lane coalesced (1 KB/instr)", f.fp);
		run<double2>(nm, buf, f.per_wg, 1, 1024, f.shared, ncu, mhz, out);
		// 8 lanes per line, 8 lines per instruction (the coupling kernel's gather)
		snprintf(nm, sizeof nm, "%s, 16 B/lane, 8 lines of 8 lanes, rows 103 KB apart", f.fp);
		if (f.per_wg >= ((size_t)2 << 20)) run<double2>(nm, buf, f.per_wg, 1, 1024, f.shared, ncu, mhz, out);
		// every lane its own line
		snprintf(nm, sizeof nm, "%s, 4 B/lane, 64 lines/instr", f.fp);
		run<uint32_t>(nm, buf, f.per_wg, 32 + 1, 64 * 33, f.shared, ncu, mhz, out);
		snprintf(nm, sizeof nm, "%s, 8 B/lane, 64 lines/instr", f.fp);
		run<double>(nm, buf, f.per_wg, 16 + 1, 64 * 17, f.shared, ncu, mhz, out);
		snprintf(nm, sizeof nm, "%s, 8 B/lane, 8 lines/instr (8 lanes per 64 B)", f.fp);
		run<double>(nm, buf, f.per_wg, 2, 64 * 2 + 16, f.shared, ncu, mhz, out);
	}
	return 0;
}
