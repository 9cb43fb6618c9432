// Calibration of HBM read counters vs access width on gfx950: streams a 4 GiB buffer with
// 4 / 8 / 16 bytes per lane (consecutive lanes -> consecutive addresses) and a 12-byte "CSR-like"
// pair of streams.  Run under rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename V> __global__ __launch_bounds__(256) void k_stream(const V* __restrict__ p, int64_t n, double* out)
{
	double s = 0;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		V v = p[i];
		const unsigned* w = (const unsigned*)&v;
		for (unsigned k = 0; k < sizeof(V) / 4; k++) s += w[k];
	}
	if (s == 1.2345) out[0] = s;
}
// one wave reads 64 consecutive elements per step, steps contiguous per wave (like one slice)
template <typename V> __global__ __launch_bounds__(256) void k_stream_wave_contig(const V* __restrict__ p, int64_t n, int per_wave_steps, double* out)
{
	double s = 0;
	const int lane = threadIdx.x & 63;
	const int64_t nw = (int64_t)gridDim.x * 4;
	const int64_t chunk = (int64_t)per_wave_steps * 64;
	for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c * chunk < n; c += nw) {
		const int64_t base = c * chunk + 3; // misaligned start like a slice
		for (int k = 0; k < per_wave_steps; k++) {
			const int64_t i = base + (int64_t)k * 64 + lane;
			if (i < n) {
				V v = p[i];
				const unsigned* w = (const unsigned*)&v;
				for (unsigned q = 0; q < sizeof(V) / 4; q++) s += w[q];
			}
		}
	}
	if (s == 1.2345) out[0] = s;
}
int main()
{
	const size_t bytes = size_t(4) << 30;
	void* buf;
	double* out;
	hipMalloc(&buf, bytes);
	hipMalloc(&out, 8);
	hipMemset(buf, 1, bytes);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float ms;
#define RUN(name, call)                                                                 \
	call; hipDeviceSynchronize(); hipEventRecord(a); call; hipEventRecord(b); hipEventSynchronize(b); \
	hipEventElapsedTime(&ms, a, b); printf("%-28s %.3f ms  %.1f GB/s\n", name, ms, bytes / 1e6 / ms);
	RUN("stream4", (k_stream<unsigned><<<4096, 256>>>((const unsigned*)buf, bytes / 4, out)));
	RUN("stream8", (k_stream<double><<<4096, 256>>>((const double*)buf, bytes / 8, out)));
	RUN("stream16", (k_stream<double2><<<4096, 256>>>((const double2*)buf, bytes / 16, out)));
	RUN("wavecontig4", (k_stream_wave_contig<unsigned><<<4096, 256>>>((const unsigned*)buf, bytes / 4, 35, out)));
	RUN("wavecontig8", (k_stream_wave_contig<double><<<4096, 256>>>((const double*)buf, bytes / 8, 35, out)));
	RUN("wavecontig16", (k_stream_wave_contig<double2><<<4096, 256>>>((const double2*)buf, bytes / 16, 35, out)));
	return 0;
}
