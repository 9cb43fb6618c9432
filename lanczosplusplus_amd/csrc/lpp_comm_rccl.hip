// lpp_comm_rccl.hip -- lpp_comm over RCCL (see include/lpp_comm_rccl.h).  Built into its own library so that
// liblpp_engine.so keeps linking nothing but the HIP runtime.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpp_comm_rccl.h"

namespace {
thread_local std::string g_err;
lpp_status fail(lpp_status code, const std::string& msg)
{
	g_err = msg;
	return code;
}
#define RT(expr)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (expr);                                                                                        \
		if (e_ != hipSuccess) return fail(LPP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
	} while (0)
#define NC(expr)                                                                                                       \
	do {                                                                                                               \
		ncclResult_t r_ = (expr);                                                                                      \
		if (r_ != ncclSuccess) return fail(LPP_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));          \
	} while (0)
} // namespace

struct lpp_rccl_comm {
	lpp_comm c {};
	ncclComm_t nccl = nullptr;
	hipStream_t compute = nullptr; // the engine's stream
	hipStream_t side = nullptr; // all-gather / exchange run here so that the local product overlaps them
	// one (ready, done) event pair per collective that runs on the side stream: [0] all-gather, [1] exchange 0, [2] exchange 1.
	// A shared pair is correct only while begin/end strictly alternate; with pairs of their own a begin of one collective can
	// never re-record an event another collective's end has still to wait for.
	hipEvent_t ev_ready[3] = { nullptr, nullptr, nullptr }, ev_done[3] = { nullptr, nullptr, nullptr };
	int ncomp = 1;
	size_t n_send = 0, n_gath = 0;
};

namespace {
// callbacks: int32 (*)(void* ctx, ...) returning 0 on success
int32_t cb_allgather_begin(void* ctx)
{
	lpp_rccl_comm* m = (lpp_rccl_comm*)ctx;
	// the slice was written on the compute stream, and the last readers of gath_buf (the previous step's remote-column
	// product) were enqueued there too: the side stream starts behind both
	if (hipEventRecord(m->ev_ready[0], m->compute) != hipSuccess) return 1;
	if (hipStreamWaitEvent(m->side, m->ev_ready[0], 0) != hipSuccess) return 1;
	if (ncclAllGather(m->c.send_buf, m->c.gath_buf, (size_t)m->c.shard_stride * m->ncomp, ncclDouble, m->nccl, m->side) != ncclSuccess) return 1;
	return hipEventRecord(m->ev_done[0], m->side) == hipSuccess ? 0 : 1;
}
int32_t cb_allgather_end(void* ctx)
{
	lpp_rccl_comm* m = (lpp_rccl_comm*)ctx;
	return hipStreamWaitEvent(m->compute, m->ev_done[0], 0) == hipSuccess ? 0 : 1;
}
int32_t cb_allreduce(void* ctx, int32_t offset, int32_t count)
{
	lpp_rccl_comm* m = (lpp_rccl_comm*)ctx;
	if (offset < 0 || count < 0 || offset + count > m->c.red_len) return 1;
	return ncclAllReduce(m->c.red_buf + offset, m->c.red_buf + offset, (size_t)count, ncclDouble, ncclSum, m->nccl, m->compute) == ncclSuccess ? 0 : 1;
}
int32_t cb_exchange_begin(void* ctx, int32_t which)
{
	lpp_rccl_comm* m = (lpp_rccl_comm*)ctx;
	const double* src = (const double*)(which == 0 ? m->c.send_buf : m->c.send2_buf);
	double* dst = (double*)(which == 0 ? m->c.gath_buf : m->c.recv2_buf);
	const size_t n = (size_t)m->c.xchg_chunk * m->ncomp;
	if (which != 0 && which != 1) return 1;
	hipEvent_t ready = m->ev_ready[1 + which], done = m->ev_done[1 + which];
	if (hipEventRecord(ready, m->compute) != hipSuccess) return 1;
	if (hipStreamWaitEvent(m->side, ready, 0) != hipSuccess) return 1;
	// all-to-all of nranks equal chunks: chunk p goes to rank p, chunk q comes from rank q (xGMI is point to point:
	// every pair has its own link, a grouped send/recv uses them all at once)
	if (ncclGroupStart() != ncclSuccess) return 1;
	bool ok = true;
	for (int p = 0; p < m->c.nranks; p++) {
		ok = ok && ncclSend(src + (size_t)p * n, n, ncclDouble, p, m->nccl, m->side) == ncclSuccess;
		ok = ok && ncclRecv(dst + (size_t)p * n, n, ncclDouble, p, m->nccl, m->side) == ncclSuccess;
	}
	if (ncclGroupEnd() != ncclSuccess || !ok) return 1;
	return hipEventRecord(done, m->side) == hipSuccess ? 0 : 1;
}
int32_t cb_exchange_end(void* ctx, int32_t which)
{
	lpp_rccl_comm* m = (lpp_rccl_comm*)ctx;
	if (which != 0 && which != 1) return 1;
	return hipStreamWaitEvent(m->compute, m->ev_done[1 + which], 0) == hipSuccess ? 0 : 1;
}
} // namespace

extern "C" {

const char* lpp_rccl_last_error(void) { return g_err.c_str(); }

lpp_status lpp_rccl_unique_id(void* id128)
{
	if (!id128) return fail(LPP_ERR_INVALID, "lpp_rccl_unique_id: null");
	static_assert(sizeof(ncclUniqueId) == LPP_RCCL_ID_BYTES, "id size");
	ncclUniqueId id;
	NC(ncclGetUniqueId(&id));
	std::memcpy(id128, &id, sizeof(id));
	return LPP_OK;
}

lpp_status lpp_rccl_comm_create(lpp_rccl_comm** out, int32_t rank, int32_t nranks, const void* id128, int32_t device, void* stream,
                                int64_t shard_stride, int32_t max_steps, int32_t is_complex, int64_t xchg_chunk)
{
	if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks || shard_stride <= 0 || max_steps < 1 || xchg_chunk < 0)
		return fail(LPP_ERR_INVALID, "lpp_rccl_comm_create: bad argument");
	RT(hipSetDevice(device));
	lpp_rccl_comm* m = new lpp_rccl_comm();
	m->ncomp = is_complex ? 2 : 1;
	m->compute = (hipStream_t)stream;
	ncclUniqueId id;
	std::memcpy(&id, id128, sizeof(id));
	ncclResult_t r = ncclCommInitRank(&m->nccl, nranks, id, rank);
	if (r != ncclSuccess) {
		delete m;
		return fail(LPP_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
	}
	hipError_t he = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking);
	for (int k = 0; k < 3; k++) {
		if (he == hipSuccess) he = hipEventCreateWithFlags(&m->ev_ready[k], hipEventDisableTiming);
		if (he == hipSuccess) he = hipEventCreateWithFlags(&m->ev_done[k], hipEventDisableTiming);
	}
	lpp_comm& c = m->c;
	c.rank = rank;
	c.nranks = nranks;
	c.ctx = m;
	c.shard_stride = shard_stride;
	c.red_len = 6 * (max_steps + 2) + 8;
	c.xchg_chunk = xchg_chunk;
	m->n_send = (size_t)(xchg_chunk > 0 ? (int64_t)nranks * xchg_chunk : shard_stride) * m->ncomp;
	m->n_gath = (size_t)(xchg_chunk > 0 ? (int64_t)nranks * xchg_chunk : (int64_t)nranks * shard_stride) * m->ncomp;
	// +2 doubles: the BLAS-1 kernels write the slice as 16-byte pairs (hipMalloc returns 256-byte aligned memory)
	if (he == hipSuccess) he = hipMalloc(&c.send_buf, sizeof(double) * (m->n_send + 2));
	if (he == hipSuccess) he = hipMalloc(&c.gath_buf, sizeof(double) * (m->n_gath + 2));
	if (he == hipSuccess) he = hipMalloc((void**)&c.red_buf, sizeof(double) * (size_t)c.red_len);
	if (he == hipSuccess && xchg_chunk > 0) he = hipMalloc(&c.send2_buf, sizeof(double) * (m->n_send + 2));
	if (he == hipSuccess && xchg_chunk > 0) he = hipMalloc(&c.recv2_buf, sizeof(double) * (m->n_gath + 2));
	if (he == hipSuccess) he = hipMemset(c.send_buf, 0, sizeof(double) * (m->n_send + 2));
	if (he == hipSuccess) he = hipMemset(c.gath_buf, 0, sizeof(double) * (m->n_gath + 2));
	if (he == hipSuccess) he = hipMemset(c.red_buf, 0, sizeof(double) * (size_t)c.red_len);
	if (he == hipSuccess && xchg_chunk > 0) he = hipMemset(c.send2_buf, 0, sizeof(double) * (m->n_send + 2));
	if (he == hipSuccess && xchg_chunk > 0) he = hipMemset(c.recv2_buf, 0, sizeof(double) * (m->n_gath + 2));
	if (he != hipSuccess) {
		lpp_rccl_comm_destroy(m);
		return fail(he == hipErrorOutOfMemory ? LPP_ERR_NOMEM : LPP_ERR_HIP, std::string("lpp_rccl_comm_create: ") + hipGetErrorString(he));
	}
	c.allgather_begin = cb_allgather_begin;
	c.allgather_end = cb_allgather_end;
	c.allreduce_sum = cb_allreduce;
	if (xchg_chunk > 0) {
		c.exchange_begin = cb_exchange_begin;
		c.exchange_end = cb_exchange_end;
	}
	*out = m;
	return LPP_OK;
}

const lpp_comm* lpp_rccl_comm_get(lpp_rccl_comm* c) { return c ? &c->c : nullptr; }

lpp_status lpp_rccl_comm_destroy(lpp_rccl_comm* m)
{
	if (!m) return LPP_OK;
	// Only what this object owns is touched: the engine's stream may already be gone (an engine that owns its stream destroys
	// it in lpp_engine_destroy).  Everything the compute stream still has to do with these buffers sits in front of the last
	// `done` event's wait, or was drained by lpp_engine_destroy / lpp_engine_sync -- close the communicator after one of them.
	if (m->side) (void)hipStreamSynchronize(m->side);
	for (int k = 0; k < 3; k++)
		if (m->ev_done[k]) (void)hipEventSynchronize(m->ev_done[k]);
	for (void* p : { m->c.send_buf, m->c.gath_buf, (void*)m->c.red_buf, m->c.send2_buf, m->c.recv2_buf })
		if (p) (void)hipFree(p);
	for (int k = 0; k < 3; k++) {
		if (m->ev_ready[k]) (void)hipEventDestroy(m->ev_ready[k]);
		if (m->ev_done[k]) (void)hipEventDestroy(m->ev_done[k]);
	}
	if (m->side) (void)hipStreamDestroy(m->side);
	if (m->nccl) (void)ncclCommDestroy(m->nccl);
	delete m;
	return LPP_OK;
}

lpp_status lpp_rccl_comm_selftest(lpp_rccl_comm* m)
{
	if (!m) return fail(LPP_ERR_INVALID, "lpp_rccl_comm_selftest: null");
	const lpp_comm& c = m->c;
	const int P = c.nranks, r = c.rank;
	// all-gather (or exchange 0): element k of rank q's slice = q*1e6 + k
	std::vector<double> h(m->n_send), g(m->n_gath);
	const size_t chunk = (size_t)c.xchg_chunk * m->ncomp;
	for (size_t k = 0; k < m->n_send; k++) h[k] = (c.xchg_chunk > 0) ? (double)r * 1e6 + (double)(k / chunk) * 1e3 + (double)(k % chunk) : (double)r * 1e6 + (double)k;
	RT(hipMemcpyAsync(c.send_buf, h.data(), sizeof(double) * m->n_send, hipMemcpyHostToDevice, m->compute));
	if (c.xchg_chunk > 0) {
		if (c.exchange_begin(c.ctx, 0) != 0 || c.exchange_end(c.ctx, 0) != 0) return fail(LPP_ERR_COMM, "selftest: exchange failed");
	} else {
		if (c.allgather_begin(c.ctx) != 0 || c.allgather_end(c.ctx) != 0) return fail(LPP_ERR_COMM, "selftest: all-gather failed");
	}
	RT(hipMemcpyAsync(g.data(), c.gath_buf, sizeof(double) * m->n_gath, hipMemcpyDeviceToHost, m->compute));
	RT(hipStreamSynchronize(m->compute));
	for (int q = 0; q < P; q++) {
		if (c.xchg_chunk > 0) {
			for (size_t k = 0; k < chunk; k++) // chunk q of the destination = chunk r of rank q's source
				if (g[(size_t)q * chunk + k] != (double)q * 1e6 + (double)r * 1e3 + (double)k) return fail(LPP_ERR_COMM, "selftest: exchange delivered wrong data");
		} else {
			const size_t n = (size_t)c.shard_stride * m->ncomp;
			for (size_t k = 0; k < n; k++)
				if (g[(size_t)q * n + k] != (double)q * 1e6 + (double)k) return fail(LPP_ERR_COMM, "selftest: all-gather delivered wrong data");
		}
	}
	// all-reduce of three scalars in the middle of the buffer
	const double v[3] = { 1.0 + r, 0.5, -2.0 * (r + 1) };
	RT(hipMemcpyAsync(c.red_buf + 5, v, sizeof(v), hipMemcpyHostToDevice, m->compute));
	if (c.allreduce_sum(c.ctx, 5, 3) != 0) return fail(LPP_ERR_COMM, "selftest: all-reduce failed");
	double w[3];
	RT(hipMemcpyAsync(w, c.red_buf + 5, sizeof(w), hipMemcpyDeviceToHost, m->compute));
	RT(hipStreamSynchronize(m->compute));
	const double e0 = P + 0.5 * P * (P - 1), e1 = 0.5 * P, e2 = -2.0 * (0.5 * P * (P + 1));
	if (std::fabs(w[0] - e0) > 1e-12 || std::fabs(w[1] - e1) > 1e-12 || std::fabs(w[2] - e2) > 1e-12) return fail(LPP_ERR_COMM, "selftest: all-reduce summed wrongly");
	RT(hipMemsetAsync(c.send_buf, 0, sizeof(double) * m->n_send, m->compute));
	RT(hipMemsetAsync(c.gath_buf, 0, sizeof(double) * m->n_gath, m->compute));
	RT(hipMemsetAsync(c.red_buf, 0, sizeof(double) * (size_t)c.red_len, m->compute));
	RT(hipStreamSynchronize(m->compute));
	return LPP_OK;
}

} // extern "C"
