// lpp_lanczos.hip -- the device-resident Lanczos loop (A2/A3 of SURVEY 8(a)).
//
// Restates LanczosSolver::computeAllStatesBelow / decomposition [PsimagLite] as called from
// reference src/Engine/Engine.h:626 and :478.  Per step j (all on the GPU, no host sync):
//   x += H y_j                         (k_spmv_*; fused partial of a_j = Re<y_j|x>)
//   x -= a_j y_j ;  b_j^2 = |x|^2      (k_axpy_nrm)          [+ blocked CGS2 when reortho]
//   (y_{j+1}, x) <- (x/b_j, -b_j y_j)  (k_swap_scale; y_{j+1} lands directly in the Krylov basis)
// The host diagonalises the (j+1)x(j+1) tridiagonal matrix `check_lag` steps behind the stream
// and stops at the first step where |E_j - E_{j-1}| < eps with j >= min_steps, exactly the
// reference's stopping rule; run-ahead steps are discarded.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "lpp_engine_impl.h"

using namespace lpp;

static int blas_blocks(int64_t n2)
{
	int64_t b = (n2 + kBlock - 1) / kBlock;
	return (int)std::max<int64_t>(1, std::min<int64_t>(b, 2048));
}

// the 2-read 1-write pass of the scale-free recurrence: 4 pairs per lane in flight, up to 16384 blocks.  Measured with
// scripts/experiments/calib_axpy.hip (non-temporal accesses): 2048 blocks 4.96 TB/s, 16384 blocks 5.52 TB/s on 1.33 GB vectors;
// 5.18 -> 6.91 TB/s on the 0.32 GB vectors of the L = 28 Heisenberg chain
// The local part of a product on the transposition exchange runs in two launches, one beside each all-to-all.  One workgroup per CU
// walks the blocks in rounds of num_cus: cut at a whole number of rounds, or the two launches together take a round more than one
// would (measured at 8 ranks of config 2, 1609 blocks: 2 x 805 = 4 + 4 rounds of 256 against 3 + 4 with the cut at 768).
static int64_t split_blocks(int64_t nid, int num_cus)
{
	const int64_t half = (nid + 1) / 2;
	if (num_cus <= 0 || half < num_cus) return half;
	return half / num_cus * num_cus;
}

static int axpy_blocks(int64_t n2)
{
	const int64_t b = (n2 + 4 * kBlock - 1) / (4 * kBlock);
	return (int)std::max<int64_t>(1, std::min<int64_t>(b, 16384));
}

lpp_status lpp_engine::adopt_comm(const lpp_comm* c)
{
	if (c->nranks < 1 || c->rank < 0 || c->rank >= c->nranks) return fail(LPP_ERR_INVALID, "lpp_comm: bad rank/nranks");
	if (c->nranks > 1) {
		if (!c->send_buf || !c->gath_buf || !c->red_buf || !c->allgather_begin || !c->allgather_end || !c->allreduce_sum)
			return fail(LPP_ERR_INVALID, "lpp_comm: missing buffer or callback");
		if (c->red_len < 4 * M + 8) return fail(LPP_ERR_INVALID, "lpp_comm: red_len < 4*(max_steps+2)+8");
		if (c->shard_stride <= 0) return fail(LPP_ERR_INVALID, "lpp_comm: shard_stride <= 0");
		// the BLAS-1 kernels store the slice for the next exchange as 16-byte pairs (the odd tail element as a scalar)
		if (((uintptr_t)c->send_buf & 15) != 0 || ((uintptr_t)c->gath_buf & 15) != 0)
			return fail(LPP_ERR_INVALID, "lpp_comm: send_buf / gath_buf must be 16-byte aligned");
	}
	comm = *c;
	has_comm = true;
	tx = false;
	if (c->nranks > 1)
		bind_scalars(c->red_buf);
	else
		bind_scalars(scal_own);
	return LPP_OK;
}

void lpp_engine::collect_spmv_times()
{
	if (spmv_events_used == 0) return;
	(void)hipStreamSynchronize(stream);
	for (size_t i = 0; i < spmv_events_used; i++) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, spmv_events[i].first, spmv_events[i].second) == hipSuccess) {
			if (spmv_event_cols.size() > i && spmv_event_cols[i] > 0) { // a blocked Gram-Schmidt bracket
				stats.reortho_ms_total += ms;
				stats.reortho_calls += 1;
				stats.reortho_columns += spmv_event_cols[i];
			} else {
				stats.spmv_ms_total += ms;
				stats.spmv_launches += 1;
			}
		}
	}
	spmv_events_used = 0;
}

namespace {

inline bool multi(const lpp_engine* e) { return e->has_comm && e->comm.nranks > 1; }

lpp_status comm_allreduce(lpp_engine* e, int offset, int count)
{
	if (!multi(e)) return LPP_OK;
	if (e->comm.allreduce_sum(e->comm.ctx, offset, count) != 0) return fail(LPP_ERR_COMM, "allreduce_sum callback failed");
	return LPP_OK;
}

struct SpmvTimer {
	lpp_engine* e;
	bool on;
	size_t idx = 0;
	// cols > 0: the bracket times a blocked Gram-Schmidt call against that many Krylov columns (lpp_stats.reortho_*)
	explicit SpmvTimer(lpp_engine* e_, int cols = 0) : e(e_), on(e_->cfg.time_kernels != 0)
	{
		if (!on) return;
		if (e->spmv_events_used == e->spmv_events.size()) {
			hipEvent_t a = nullptr, b = nullptr;
			if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
				on = false;
				return;
			}
			e->spmv_events.emplace_back(a, b);
		}
		idx = e->spmv_events_used++;
		if (e->spmv_event_cols.size() <= idx) e->spmv_event_cols.resize(idx + 1, 0);
		e->spmv_event_cols[idx] = cols;
		(void)hipEventRecord(e->spmv_events[idx].first, e->stream);
	}
	bool stopped = false;
	void stop()
	{
		if (on && !stopped) (void)hipEventRecord(e->spmv_events[idx].second, e->stream);
		stopped = true;
	}
	// every return path closes its bracket: an event pair whose end was never recorded would fail hipEventElapsedTime later
	~SpmvTimer() { stop(); }
	SpmvTimer(const SpmvTimer&) = delete;
	SpmvTimer& operator=(const SpmvTimer&) = delete;
};

// blocked classical Gram-Schmidt, two passes, of x against Krylov columns [0, ncol)
lpp_status cgs2(lpp_engine* e, int ncol)
{
	const int nb = blas_blocks(e->n2);
	const int64_t ldv2 = e->ldv / 2;
	for (int pass = 0; pass < 2; pass++) {
		for (int p0 = 0; p0 < ncol; p0 += kPanel) {
			const int np = std::min(kPanel, ncol - p0);
			const double2* v0 = (const double2*)(e->V + (int64_t)p0 * e->ldv);
			if (e->is_complex)
				k_multi_dot<true><<<nb, kBlock, 0, e->stream>>>((const double2*)e->x, v0, ldv2, np, e->n2, e->partial);
			else
				k_multi_dot<false><<<nb, kBlock, 0, e->stream>>>((const double2*)e->x, v0, ldv2, np, e->n2, e->partial);
			k_reduce_final<<<1, kBlock, 0, e->stream>>>(e->partial, nb, 2 * kPanel, 2 * np, e->coef_dev + 2 * p0);
		}
		lpp_status st = comm_allreduce(e, e->coef_off, 2 * ncol);
		if (st != LPP_OK) return st;
		for (int p0 = 0; p0 < ncol; p0 += kPanel) {
			const int np = std::min(kPanel, ncol - p0);
			const double2* v0 = (const double2*)(e->V + (int64_t)p0 * e->ldv);
			if (e->is_complex)
				k_multi_axpy<true><<<nb, kBlock, 0, e->stream>>>((double2*)e->x, v0, ldv2, np, e->coef_dev + 2 * p0, -1.0, e->n2);
			else
				k_multi_axpy<false><<<nb, kBlock, 0, e->stream>>>((double2*)e->x, v0, ldv2, np, e->coef_dev + 2 * p0, -1.0, e->n2);
		}
	}
	return LPP_OK;
}

// enqueue Lanczos step j = e->step.  ritz (optional): nst coefficients to accumulate
// zwork_k += ritz[k] * (current Lanczos vector) during a second pass.
//
// Two forms of the same recurrence:
//  normalised (vectors kept / reortho):  x += H y_j; a_j = <y_j|x>; x -= a_j y_j; b_j = |x|; (y_{j+1}, x) <- (x/b_j, -b_j y_j)
//  scale-free (no vectors kept):         r_j = b_{j-1} y_j stays unnormalised;  w = H r_j / b_{j-1} - (b_{j-1}/b_{j-2}) r_{j-1}
//                                        is formed by the SpMV epilogue in the buffer of r_{j-1}; raw_j = <r_j|w> = a_j b_{j-1};
//                                        r_{j+1} = w - (raw_j / b_{j-1}^2) r_j; b_j = |r_{j+1}|.  No swap/scale pass.
lpp_status one_step(lpp_engine* e, const double* ritz, int nst)
{
	const int j = e->step;
	const int nb = blas_blocks(e->n2);
	hipStream_t st = e->stream;
	double* ycur = e->ycur;
	double* xcur = e->xcur;
	int np = 0;
	bool tx_pair = false;
	std::unique_ptr<SpmvTimer> pb_timer; // closes its bracket on every return path
	// product-basis layout, no vectors kept: two launches per step, the axpy of step j rides in the product of step j+1
	const bool pb_chain = e->pb.active && e->scalefree && !ritz && pb_chain_ok(e);
	// the same deferral on the transposition exchange: the update rides in the next step's pack kernel (k_pack_transpose_axpy)
	const bool pb_lazy_tx = e->pb.active && e->pb.tx && multi(e) && e->tx && e->scalefree && !ritz &&
	                        !(getenv("LPP_PB_LAZY_TX") && atoi(getenv("LPP_PB_LAZY_TX")) == 0);
	if (e->pb.pending && !pb_chain && !pb_lazy_tx) { // someone needs r_j itself: run the pass the chain left out
		pb_materialise(e, ycur, xcur, e->pb.pend_a, e->pb.pend_b2, e->partial);
		e->pb.pending = false;
	}
	if (ritz) {
		for (int k = 0; k < nst; k++)
			k_axpy_const<<<nb, kBlock, 0, st>>>((double2*)(e->zwork + (int64_t)k * e->nd_pad), (const double2*)ycur, ritz[k], e->n2);
	}
	double* a_ptr = e->ab_dev + 2 * j;
	double* b2_ptr = e->ab_dev + 2 * j + 1;
	EpiScale sc { nullptr, nullptr, 0 };
	const double* b2_prev = nullptr;
	if (e->scalefree) {
		b2_prev = (j == 0) ? e->tmp_dev : e->ab_dev + 2 * (j - 1) + 1;
		sc.b2_prev = b2_prev;
		sc.b2_prev2 = (j == 0) ? nullptr : ((j == 1) ? e->tmp_dev : e->ab_dev + 2 * (j - 2) + 1);
	}
	if (multi(e) && e->tx) {
		// Transposition exchange (stored and matrix-free engines): two all-to-alls of N/P elements instead of an
		// all-gather of N; with the matrix-free engine nothing of size N ever exists on a rank.
		//   pack -> all-to-all #1  ||  local part (diagonal/U + up-hops on the own slice)
		//   -> down-hops on the received transposed slice -> all-to-all #2 -> unpack-add (+ fused a_j partial)
		const int64_t n_up = e->kron_n_up_tx, nid = e->n_local / n_up, chunk = e->comm.xchg_chunk;
		const int nbp = (int)std::max<int64_t>(1, std::min<int64_t>((e->n_local + kBlock - 1) / kBlock, 2048));
		if (e->is_complex)
			k_pack_transpose<cplx><<<nbp, kBlock, 0, st>>>((const cplx*)ycur, (cplx*)e->comm.send_buf, nid, n_up, e->tx_peru, chunk);
		else if (pb_lazy_tx && e->pb.pending)
			k_pack_transpose_axpy<<<nbp, kBlock, 0, st>>>(ycur, xcur, e->pb.pend_a, e->pb.pend_b2, (double*)e->comm.send_buf, nid, n_up, e->tx_peru, chunk, e->pitch);
		else
			k_pack_transpose<double><<<nbp, kBlock, 0, st>>>((const double*)ycur, (double*)e->comm.send_buf, nid, n_up, e->tx_peru, chunk, e->pitch);
		if (e->pb.active) {
		// product-basis layout: both parts are the single-GPU kernels (lpp_pb.hip, "several GPUs"); each all-to-all has half of
		// the in-block part to hide behind
		const int64_t half = split_blocks(nid, e->num_cus);
		if (e->comm.exchange_begin(e->comm.ctx, 0) != 0) return fail(LPP_ERR_COMM, "exchange_begin(0) callback failed");
		{
			SpmvTimer t(e);
			pb_tx_up(e, ycur, sc, 0, half);
			t.stop();
		}
		if (e->comm.exchange_end(e->comm.ctx, 0) != 0) return fail(LPP_ERR_COMM, "exchange_end(0) callback failed");
		{
			SpmvTimer t(e);
			pb_tx_down(e, e->comm.gath_buf, e->comm.send2_buf, sc);
			t.stop();
		}
		if (e->comm.exchange_begin(e->comm.ctx, 1) != 0) return fail(LPP_ERR_COMM, "exchange_begin(1) callback failed");
		{
			SpmvTimer t(e);
			pb_tx_up(e, ycur, sc, half, nid - half);
			t.stop();
		}
		if (e->comm.exchange_end(e->comm.ctx, 1) != 0) return fail(LPP_ERR_COMM, "exchange_end(1) callback failed");
		np = pb_tx_unpack_combine(e, xcur, ycur, e->comm.recv2_buf, sc, chunk, e->partial, e->scalefree ? e->tmp_dev + 1 : nullptr);
		tx_pair = true;
	} else {
		if (e->comm.exchange_begin(e->comm.ctx, 0) != 0) return fail(LPP_ERR_COMM, "exchange_begin(0) callback failed");
		// The local part (diagonal / U + up-hops on the own slice) needs no exchange.  The matrix-free engine runs the first
		// half of its blocks beside all-to-all #1 and the second half beside all-to-all #2, so both transfers have a kernel to
		// hide behind; the stored local matrix is one launch, beside #1.
		const int64_t half = e->kron.active ? split_blocks(nid, e->num_cus) : nid;
		{
			SpmvTimer t(e); // overlaps all-to-all #1
			if (e->kron.active)
				kron_launch(e, ycur, ycur, xcur, nullptr, sc, 1, 0, half);
			else
				spmv_launch(e, e->A_loc, ycur, xcur, nullptr, nullptr, sc);
			t.stop();
		}
		if (e->comm.exchange_end(e->comm.ctx, 0) != 0) return fail(LPP_ERR_COMM, "exchange_end(0) callback failed");
		HIP_TRY(hipMemsetAsync(e->comm.send2_buf, 0, e->esz * (size_t)chunk * (size_t)e->comm.nranks, st));
		{
			EpiScale sc2 = sc;
			sc2.beta_one = 1; // wT = alpha * (down-hop part) yT into the zeroed buffer
			SpmvTimer t(e);
			if (e->kron.active)
				kron_launch(e, nullptr, e->comm.gath_buf, e->comm.send2_buf, nullptr, sc2, 2);
			else
				spmv_launch(e, e->A_rem, e->comm.gath_buf, e->comm.send2_buf, nullptr, nullptr, sc2);
			t.stop();
		}
		if (e->comm.exchange_begin(e->comm.ctx, 1) != 0) return fail(LPP_ERR_COMM, "exchange_begin(1) callback failed");
		if (half < nid) {
			SpmvTimer t(e); // overlaps all-to-all #2
			kron_launch(e, ycur, ycur, xcur, nullptr, sc, 1, half, nid - half);
			t.stop();
		}
		if (e->comm.exchange_end(e->comm.ctx, 1) != 0) return fail(LPP_ERR_COMM, "exchange_end(1) callback failed");
		if (e->is_complex)
			k_unpack_add_dot<cplx, true><<<nbp, kBlock, 0, st>>>((cplx*)xcur, (const cplx*)e->comm.recv2_buf, (const cplx*)ycur, nid, n_up, e->tx_peru, chunk, e->partial, e->scalefree ? e->tmp_dev + 1 : nullptr);
		else
			k_unpack_add_dot<double, true><<<nbp, kBlock, 0, st>>>((double*)xcur, (const double*)e->comm.recv2_buf, (const double*)ycur, nid, n_up, e->tx_peru, chunk, e->partial, e->scalefree ? e->tmp_dev + 1 : nullptr);
		np = nbp;
		tx_pair = true; // the partials come in (Re<y|x>, |x|^2) pairs
	}
	} else if (pb_chain) {
		SpmvTimer t(e);
		np = pb_launch_chain(e, ycur, xcur, e->partial, sc, e->pb.pending ? e->pb.pend_a : nullptr, e->pb.pend_b2, e->tmp_dev + 1);
		t.stop();
		tx_pair = true; // pairs (Re<r_j|w_j>, |w_j|^2)
	} else if (e->pb.active) {
		// scale-free: x is formed by pb_combine_axpy below, together with the recurrence update; the bracket then closes behind that
		// pass (pb_timer), so that the timed launches are the WHOLE step as in the chained form
		pb_timer.reset(new SpmvTimer(e));
		np = pb_launch(e, ycur, xcur, e->partial, sc, e->scalefree);
		if (!e->scalefree || multi(e)) pb_timer.reset();
	} else if (e->kron.active) {
		// matrix-free product with the all-gather: the down part needs the whole vector, so the gather completes first
		if (multi(e)) {
			if (e->comm.allgather_begin(e->comm.ctx) != 0) return fail(LPP_ERR_COMM, "allgather_begin callback failed");
			if (e->comm.allgather_end(e->comm.ctx) != 0) return fail(LPP_ERR_COMM, "allgather_end callback failed");
		}
		SpmvTimer t(e);
		np = kron_launch(e, ycur, multi(e) ? e->comm.gath_buf : ycur, xcur, e->partial, sc);
		t.stop();
	} else if (multi(e)) {
		// the slice of the current vector was written to comm.send_buf by the previous step's last kernel
		if (e->comm.allgather_begin(e->comm.ctx) != 0) return fail(LPP_ERR_COMM, "allgather_begin callback failed");
		{
			SpmvTimer t(e);
			spmv_launch(e, e->A_loc, ycur, xcur, nullptr, nullptr, sc); // local columns: overlaps the all-gather
			t.stop();
		}
		if (e->comm.allgather_end(e->comm.ctx) != 0) return fail(LPP_ERR_COMM, "allgather_end callback failed");
		{
			EpiScale sc2 = sc;
			sc2.beta_one = 1; // x already holds beta*x_old + alpha*(A_loc y)
			SpmvTimer t(e);
			np = spmv_launch(e, e->A_rem, e->comm.gath_buf, xcur, ycur, e->partial, sc2);
			t.stop();
		}
	} else {
		SpmvTimer t(e);
		np = spmv_launch(e, e->A_loc, ycur, xcur, ycur, e->partial, sc);
		t.stop();
	}
	const bool pb_sf = e->pb.active && !e->pb.tx && e->scalefree && !pb_chain; // single GPU, three-kernel form
	// transposition exchange + scale-free recurrence: a_j and b_j^2 share ONE all-reduce (k_b2_from_w)
	const bool fused_ab = pb_chain || (tx_pair && e->scalefree && !(getenv("LPP_FUSED_ALLREDUCE") && atoi(getenv("LPP_FUSED_ALLREDUCE")) == 0));
	lpp_status rc = LPP_OK;
	if (pb_sf) // the product kernels never read x: raw_j = Re<y | u + z> + beta Re<y | x_old>
		k_pb_reduce_a<<<1, kBlock, 0, st>>>(e->partial, np, e->pb.xy, sc, a_ptr);
	else if (tx_pair)
		k_reduce_final<<<1, kBlock, 0, st>>>(e->partial, np, 2, fused_ab ? 2 : 1, a_ptr); // a_ptr[0] = a_j (raw), a_ptr[1] = |w|^2
	else
		k_reduce_final<<<1, kBlock, 0, st>>>(e->partial, np, 1, 1, a_ptr);
	rc = comm_allreduce(e, e->ab_off + 2 * j, fused_ab ? 2 : 1);
	if (rc != LPP_OK) return rc;
	int nb_nrm = nb;
	if (pb_chain || (pb_lazy_tx && fused_ab)) {
		k_b2_from_w<<<1, 64, 0, st>>>(a_ptr, b2_prev, e->tmp_dev + 1);
		e->pb.pending = true; // r_{j+1} = w_j - (raw_j / b_{j-1}^2) r_j is formed by the next step's k_pb_up / pack kernel
		e->pb.pend_a = a_ptr;
		e->pb.pend_b2 = b2_prev;
	} else if (fused_ab) {
		k_b2_from_w<<<1, 64, 0, st>>>(a_ptr, b2_prev, e->tmp_dev + 1);
		k_axpy_nrm<false><<<nb, kBlock, 0, st>>>((double2*)xcur, (const double2*)ycur, a_ptr, b2_prev, nullptr, e->n2, nullptr, 0, e->nd);
	} else if (pb_sf) {
		nb_nrm = pb_combine_axpy(e, xcur, ycur, sc, a_ptr, b2_prev, e->partial);
		if (pb_timer) pb_timer->stop();
	} else if (e->scalefree) {
		// streamed accesses once the two vectors no longer fit the 256 MiB Infinity Cache (measured: +4 % there, -7 % below)
		const int stream_axpy = (size_t)e->n2 * 32 > ((size_t)256 << 20) ? 1 : 0;
		nb_nrm = stream_axpy ? axpy_blocks(e->n2) : nb;
		k_axpy_nrm<true><<<nb_nrm, kBlock, 0, st>>>((double2*)xcur, (const double2*)ycur, a_ptr, b2_prev,
		                                       (multi(e) && !e->tx) ? (double2*)e->comm.send_buf : nullptr, e->n2, e->partial, stream_axpy, e->nd);
	} else if (e->cfg.reortho) {
		k_axpy_nrm<false><<<nb, kBlock, 0, st>>>((double2*)xcur, (const double2*)ycur, a_ptr, nullptr, nullptr, e->n2, nullptr);
		{
			SpmvTimer t(e, j + 1);
			rc = cgs2(e, j + 1);
			t.stop();
		}
		if (rc != LPP_OK) return rc;
		k_dot<<<nb, kBlock, 0, st>>>((const double2*)xcur, (const double2*)xcur, e->n2, e->partial);
	} else {
		k_axpy_nrm<true><<<nb, kBlock, 0, st>>>((double2*)xcur, (const double2*)ycur, a_ptr, nullptr, nullptr, e->n2, e->partial);
	}
	if (pb_timer) {
		pb_timer.reset(); // stopped above on every path that created it
	}
	if (!fused_ab) {
		k_reduce_final<<<1, kBlock, 0, st>>>(e->partial, nb_nrm, 1, 1, b2_ptr);
		rc = comm_allreduce(e, e->ab_off + 2 * j + 1, 1);
		if (rc != LPP_OK) return rc;
	}
	if (e->scalefree) {
		// roles swap: the buffer that held r_{j-1} now holds r_{j+1}
		e->ycur = xcur;
		e->xcur = ycur;
	} else {
		double* ynext = e->saving ? e->V + (int64_t)(j + 1) * e->ldv : e->y;
		k_swap_scale<<<nb, kBlock, 0, st>>>((double2*)xcur, (const double2*)ycur, (double2*)ynext,
		                                   (multi(e) && !e->tx) ? (double2*)e->comm.send_buf : nullptr, b2_ptr, e->n2, e->nd);
		e->ycur = ynext;
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_scal + 2 * j, e->ab_dev + 2 * j, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
	if ((int)e->step_events.size() <= j) {
		const size_t old = e->step_events.size();
		e->step_events.resize(j + 1, nullptr);
		for (size_t i = old; i < e->step_events.size(); i++) HIP_TRY(hipEventCreateWithFlags(&e->step_events[i], hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(e->step_events[j], st));
	e->step = j + 1;
	e->stats.steps_enqueued = e->step;
	return LPP_OK;
}

// coefficients (a_j, b_j) of step k from the pinned mirror; in the scale-free form the device holds raw_k = a_k b_{k-1}
void host_coeffs(const lpp_engine* e, int k, double b_prev, double* a, double* b)
{
	const double b2 = e->h_scal[2 * k + 1];
	*b = std::sqrt(b2 > 0 ? b2 : 0.0);
	double av = e->h_scal[2 * k];
	if (e->scalefree && std::fabs(b_prev) >= 1e-10) av /= b_prev;
	*a = av;
}

int effective_max_steps(const lpp_engine* e)
{
	return (int)std::min<int64_t>(e->cfg.max_steps, std::max<int64_t>(e->n_global, 1));
}

lpp_status ensure_krylov(lpp_engine* e, int ncols, bool required, bool* got)
{
	*got = false;
	const int64_t ldv = (e->nd_pad + 31) & ~(int64_t)31; // 256-byte aligned columns
	if (e->V && e->vcap >= ncols && e->ldv == ldv) {
		*got = true;
		return LPP_OK;
	}
	if (e->V) {
		(void)hipFree(e->V);
		e->V = nullptr;
		e->vcap = 0;
	}
	const size_t bytes = sizeof(double) * (size_t)ldv * (size_t)ncols;
	if (!required) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes + (size_t(2) << 30) > free_b) return LPP_OK;
	}
	hipError_t err = hipMalloc(&e->V, std::max<size_t>(bytes, 16));
	if (err != hipSuccess) {
		e->V = nullptr;
		(void)hipGetLastError();
		if (required) return fail(LPP_ERR_NOMEM, "cannot allocate the on-device Krylov basis (reortho / save_vectors need max_steps+1 vectors)");
		return LPP_OK;
	}
	e->ldv = ldv;
	e->vcap = ncols;
	*got = true;
	return LPP_OK;
}

lpp_status begin_run(lpp_engine* e, const void* init, bool want_save)
{
	if (!e->has_matrix()) return fail(LPP_ERR_STATE, "no matrix: call lpp_engine_set_csr / lpp_engine_assemble_* first");
	HIP_TRY(hipSetDevice(e->cfg.device));
	if (e->n_global <= 0) return fail(LPP_ERR_INVALID, "empty matrix");
	const int maxs = effective_max_steps(e);
	bool need = e->cfg.reortho != 0 || e->cfg.save_vectors == 1;
	bool got = false;
	if (need || (want_save && e->cfg.save_vectors != 0)) {
		lpp_status st = ensure_krylov(e, maxs + 1, need, &got);
		if (st != LPP_OK) return st;
	}
	e->saving = got;
	e->scalefree = !e->saving && !e->cfg.reortho && getenv("LPP_NO_SCALE_FREE") == nullptr;
	e->step = 0;
	hipStream_t st = e->stream;
	const int nb = blas_blocks(e->n2);
	// start vector -> x (scratch)
	HIP_TRY(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)e->nd_pad, st));
	if (init) {
		lpp_status rc0 = vec_from_host(e, e->x, init);
		if (rc0 != LPP_OK) return rc0;
	} else {
		vec_fill_random(e, e->x, e->cfg.seed);
	}
	k_dot<<<nb, kBlock, 0, st>>>((const double2*)e->x, (const double2*)e->x, e->n2, e->partial);
	k_reduce_final<<<1, kBlock, 0, st>>>(e->partial, nb, 1, 1, e->tmp_dev);
	lpp_status rc = comm_allreduce(e, e->tmp_off, 1);
	if (rc != LPP_OK) return rc;
	HIP_TRY(hipMemcpyAsync(e->h_scal + 2 * e->M, e->tmp_dev, sizeof(double), hipMemcpyDeviceToHost, st)); // |init|^2 = b_{-1}^2
	if (e->pb.active) HIP_TRY(hipMemsetAsync(e->pb.xy, 0, sizeof(double) * 2, st)); // <y | x_old> of step 0: x_old = 0
	e->pb.pending = false;
	HIP_TRY(hipMemsetAsync(e->tmp_dev + 1, 0, sizeof(double), st)); // shift of the derived norm (k_b2_from_w): none at step 0
	if (e->scalefree) {
		// r_0 = init stays unnormalised in e->x; e->y is the (zero) buffer of r_{-1}
		e->ycur = e->x;
		e->xcur = e->y;
		HIP_TRY(hipMemsetAsync(e->y, 0, sizeof(double) * (size_t)e->nd_pad, st));
		if (e->pb.active && e->pb.dval && !e->pb.tx) { // plain-stream diagonal: <r_0 | D r_0> for step 0's a_j (the combine pass carries it from then on)
			k_pb_dq<<<nb, kBlock, 0, st>>>((const double2*)e->x, (const double2*)e->pb.dval, e->n2, e->partial);
			k_reduce_final<<<1, kBlock, 0, st>>>(e->partial, nb, 1, 1, e->pb.xy + 1);
		}
		if (multi(e) && !e->tx) HIP_TRY(hipMemcpyAsync(e->comm.send_buf, e->x, sizeof(double) * (size_t)e->nd, hipMemcpyDeviceToDevice, st));
	} else {
		e->ycur = e->saving ? e->V : e->y;
		e->xcur = e->x;
		k_scale_copy<<<nb, kBlock, 0, st>>>((double2*)e->ycur, (multi(e) && !e->tx) ? (double2*)e->comm.send_buf : nullptr, (const double2*)e->x, e->tmp_dev, e->n2, e->nd);
		HIP_TRY(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)e->nd_pad, st));
	}
	HIP_TRY(hipGetLastError());
	e->active = true;
	e->spmv_events_used = 0;
	std::memset(&e->stats, 0, sizeof(e->stats));
	e->stats.vectors_saved = e->saving ? 1 : 0;
	return LPP_OK;
}

struct SolveResult {
	int steps = 0;
	bool converged = false;
	std::vector<double> a, b;
};

// run the recurrence with the lagged convergence test; fills res
lpp_status run_recurrence(lpp_engine* e, SolveResult& res)
{
	const int maxs = effective_max_steps(e);
	const int lag = e->cfg.check_lag;
	res.a.assign(maxs, 0.0);
	res.b.assign(maxs, 0.0);
	double eold = 100.0;
	int checked = 0;
	int final_steps = -1;
	auto check = [&](int k) -> lpp_status {
		HIP_TRY(hipEventSynchronize(e->step_events[k]));
		const double b_prev = (k == 0) ? std::sqrt(std::max(e->h_scal[2 * e->M], 0.0)) : res.b[k - 1];
		host_coeffs(e, k, b_prev, &res.a[k], &res.b[k]);
		if (!std::isfinite(res.a[k]) || !std::isfinite(res.b[k])) return fail(LPP_ERR_NOCONV, "Lanczos produced a non-finite coefficient");
		const double enew = tridiag_kth(k + 1, res.a.data(), res.b.data(), 0);
		if (e->cfg.eps > 0) {
			const bool exitFlag = std::fabs(enew - eold) < e->cfg.eps;
			if (exitFlag && (e->n_global <= 4 || k >= e->cfg.min_steps)) {
				final_steps = k + 1;
				res.converged = true;
			}
		}
		if (final_steps < 0 && std::fabs(res.b[k]) < 1e-10) { // invariant subspace exhausted
			final_steps = k + 1;
			res.converged = true;
		}
		eold = enew;
		return LPP_OK;
	};
	for (int j = 0; j < maxs && final_steps < 0; j++) {
		lpp_status st = one_step(e, nullptr, 0);
		if (st != LPP_OK) return st;
		while (final_steps < 0 && checked <= j - lag) {
			st = check(checked++);
			if (st != LPP_OK) return st;
		}
	}
	while (final_steps < 0 && checked < e->step) {
		lpp_status st = check(checked++);
		if (st != LPP_OK) return st;
	}
	if (final_steps < 0) final_steps = e->step;
	HIP_TRY(hipStreamSynchronize(e->stream));
	res.steps = final_steps;
	res.a.resize(final_steps);
	res.b.resize(final_steps);
	return LPP_OK;
}

} // namespace

extern "C" {

lpp_status lpp_engine_set_solver(lpp_engine* e, int32_t max_steps, int32_t min_steps, double eps, int32_t reortho, int32_t save_vectors)
{
	if (!e || max_steps < 1 || min_steps < 0) return fail(LPP_ERR_INVALID, "lpp_engine_set_solver: bad argument");
	if (e->active) return fail(LPP_ERR_STATE, "lpp_engine_set_solver: a Lanczos run is active");
	HIP_TRY(hipSetDevice(e->cfg.device));
	const int M = max_steps + 2;
	if (M != e->M) {
		if (multi(e) && e->comm.red_len < 4 * M + 8) return fail(LPP_ERR_INVALID, "lpp_engine_set_solver: comm.red_len < 4*(max_steps+2)+8");
		HIP_TRY(hipStreamSynchronize(e->stream));
		double *scal = nullptr, *hs = nullptr;
		HIP_TRY_MEM(hipMalloc(&scal, sizeof(double) * (size_t)(6 * M + 8)));
		if (hipHostMalloc(&hs, sizeof(double) * (size_t)(2 * M + 2), hipHostMallocDefault) != hipSuccess) {
			(void)hipFree(scal);
			return fail(LPP_ERR_NOMEM, "lpp_engine_set_solver: pinned allocation failed");
		}
		if (e->scal_own) (void)hipFree(e->scal_own);
		if (e->h_scal) (void)hipHostFree(e->h_scal);
		e->scal_own = scal;
		e->h_scal = hs;
		e->M = M;
		e->bind_scalars(multi(e) ? e->comm.red_buf : e->scal_own);
	}
	e->cfg.max_steps = max_steps;
	e->cfg.min_steps = min_steps;
	e->cfg.eps = eps;
	e->cfg.reortho = reortho ? 1 : 0;
	e->cfg.save_vectors = save_vectors;
	return LPP_OK;
}

lpp_status lpp_engine_lanczos_begin(lpp_engine* e, const void* init)
{
	if (!e) return fail(LPP_ERR_INVALID, "lpp_engine_lanczos_begin: null engine");
	return begin_run(e, init, false);
}

lpp_status lpp_engine_lanczos_step(lpp_engine* e, int32_t nsteps)
{
	if (!e || nsteps < 0) return fail(LPP_ERR_INVALID, "lpp_engine_lanczos_step: bad argument");
	if (!e->active) return fail(LPP_ERR_STATE, "lpp_engine_lanczos_step: call lpp_engine_lanczos_begin first");
	HIP_TRY(hipSetDevice(e->cfg.device));
	for (int i = 0; i < nsteps; i++) {
		if (e->step >= effective_max_steps(e)) return fail(LPP_ERR_STATE, "lpp_engine_lanczos_step: max_steps reached");
		lpp_status st = one_step(e, nullptr, 0);
		if (st != LPP_OK) return st;
	}
	return LPP_OK;
}

lpp_status lpp_engine_lanczos_coeffs(lpp_engine* e, int32_t* steps, double* a, double* b)
{
	if (!e || !steps) return fail(LPP_ERR_INVALID, "lpp_engine_lanczos_coeffs: bad argument");
	HIP_TRY(hipSetDevice(e->cfg.device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	*steps = e->step;
	double b_prev = std::sqrt(std::max(e->h_scal[2 * e->M], 0.0));
	for (int j = 0; j < e->step; j++) {
		double av = 0, bv = 0;
		host_coeffs(e, j, b_prev, &av, &bv);
		if (a) a[j] = av;
		if (b) b[j] = bv;
		b_prev = bv;
	}
	return LPP_OK;
}

lpp_status lpp_engine_lanczos(lpp_engine* e, const void* init, int32_t nstates, double* eigs, void* ritz_vectors, lpp_stats* stats)
{
	if (!e || nstates < 1 || !eigs) return fail(LPP_ERR_INVALID, "lpp_engine_lanczos: bad argument");
	const auto t0 = std::chrono::steady_clock::now();
	lpp_status st = begin_run(e, init, ritz_vectors != nullptr);
	if (st != LPP_OK) return st;
	SolveResult res;
	st = run_recurrence(e, res);
	e->active = false;
	if (st != LPP_OK) return st;
	const int steps = res.steps;
	if (steps < nstates) return fail(LPP_ERR_NOCONV, "Lanczos: fewer steps than requested states");
	std::vector<double> w(nstates), S;
	if (ritz_vectors) {
		S.resize((size_t)steps * nstates);
		st = lpp_tridiag_lowest(steps, res.a.data(), res.b.data(), nstates, w.data(), S.data());
	} else {
		st = lpp_tridiag_lowest(steps, res.a.data(), res.b.data(), nstates, w.data(), nullptr);
	}
	if (st != LPP_OK) return st;
	for (int k = 0; k < nstates; k++) eigs[k] = w[k];
	const int steps_enq = e->stats.steps_enqueued;
	if (ritz_vectors) {
		const int nb = blas_blocks(e->n2);
		if (e->saving) {
			// z_k = sum_j S(j,k) v_j from the on-device Krylov basis (x is free after the run)
			std::vector<double> coef(2 * (size_t)steps);
			for (int k = 0; k < nstates; k++) {
				for (int j = 0; j < steps; j++) {
					coef[2 * j] = S[(size_t)j * nstates + k];
					coef[2 * j + 1] = 0.0;
				}
				HIP_TRY(hipMemcpyAsync(e->coef_dev, coef.data(), sizeof(double) * 2 * (size_t)steps, hipMemcpyHostToDevice, e->stream));
				HIP_TRY(hipMemsetAsync(e->x, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
				for (int p0 = 0; p0 < steps; p0 += kPanel) {
					const int np = std::min(kPanel, steps - p0);
					const double2* v0 = (const double2*)(e->V + (int64_t)p0 * e->ldv);
					if (e->is_complex)
						k_multi_axpy<true><<<nb, kBlock, 0, e->stream>>>((double2*)e->x, v0, e->ldv / 2, np, e->coef_dev + 2 * p0, 1.0, e->n2);
					else
						k_multi_axpy<false><<<nb, kBlock, 0, e->stream>>>((double2*)e->x, v0, e->ldv / 2, np, e->coef_dev + 2 * p0, 1.0, e->n2);
				}
				HIP_TRY(hipGetLastError());
				st = vec_to_host(e, (char*)ritz_vectors + (size_t)k * e->esz * (size_t)e->n_local, e->x);
				if (st != LPP_OK) return st;
				HIP_TRY(hipStreamSynchronize(e->stream));
			}
		} else {
			// second pass: replay the recurrence (bitwise identical kernels) and accumulate z_k += S(j,k) y_j
			if (e->zwork) (void)hipFree(e->zwork);
			e->zwork = nullptr;
			HIP_TRY_MEM(hipMalloc(&e->zwork, sizeof(double) * (size_t)e->nd_pad * (size_t)nstates));
			lpp_stats keep = e->stats;
			st = begin_run(e, init, false);
			if (st != LPP_OK) return st;
			HIP_TRY(hipMemsetAsync(e->zwork, 0, sizeof(double) * (size_t)e->nd_pad * (size_t)nstates, e->stream));
			std::vector<double> coefk(nstates);
			const double binit = std::sqrt(std::max(e->h_scal[2 * e->M], 0.0));
			for (int j = 0; j < steps; j++) {
				// scale-free form: the vector in hand is r_j = b_{j-1} y_j (b_{-1} = |init|)
				const double bprev = (j == 0) ? binit : res.b[j - 1];
				const double sc = (e->scalefree && std::fabs(bprev) >= 1e-10) ? 1.0 / bprev : 1.0;
				for (int k = 0; k < nstates; k++) coefk[k] = S[(size_t)j * nstates + k] * sc;
				st = one_step(e, coefk.data(), nstates);
				if (st != LPP_OK) return st;
			}
			HIP_TRY(hipStreamSynchronize(e->stream));
			e->active = false;
			e->collect_spmv_times();
			keep.spmv_ms_total += e->stats.spmv_ms_total;
			keep.spmv_launches += e->stats.spmv_launches;
			e->stats = keep;
			for (int k = 0; k < nstates; k++) {
				st = vec_to_host(e, (char*)ritz_vectors + (size_t)k * e->esz * (size_t)e->n_local, e->zwork + (int64_t)k * e->nd_pad);
				if (st != LPP_OK) return st;
			}
			HIP_TRY(hipStreamSynchronize(e->stream));
			(void)hipFree(e->zwork);
			e->zwork = nullptr;
		}
	}
	e->collect_spmv_times();
	e->stats.steps = steps;
	e->stats.steps_enqueued = steps_enq;
	e->stats.converged = res.converged ? 1 : 0;
	e->stats.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (stats) lpp_engine_get_stats(e, stats);
	return LPP_OK;
}

lpp_status lpp_engine_decomposition(lpp_engine* e, const void* init, int32_t* nsteps, double* a, double* b, lpp_stats* stats)
{
	if (!e || !nsteps || !a || !b) return fail(LPP_ERR_INVALID, "lpp_engine_decomposition: bad argument");
	const auto t0 = std::chrono::steady_clock::now();
	lpp_status st = begin_run(e, init, false);
	if (st != LPP_OK) return st;
	SolveResult res;
	st = run_recurrence(e, res);
	e->active = false;
	if (st != LPP_OK) return st;
	*nsteps = res.steps;
	for (int j = 0; j < res.steps; j++) {
		a[j] = res.a[j];
		b[j] = res.b[j];
	}
	const int steps_enq = e->stats.steps_enqueued;
	e->collect_spmv_times();
	e->stats.steps = res.steps;
	e->stats.steps_enqueued = steps_enq;
	e->stats.converged = res.converged ? 1 : 0;
	e->stats.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (stats) lpp_engine_get_stats(e, stats);
	return LPP_OK;
}

} // extern "C"
