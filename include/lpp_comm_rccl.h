/*
 * lpp_comm_rccl.h -- a ready-made lpp_comm (include/lpp_engine.h) over RCCL, for hosts that are not Python:
 * one process per GPU, collectives issued from C on HIP streams (liblpp_comm_rccl.so, links librccl).
 *
 * The reference has no communication layer at all ("Distributed communication backend: None"); this is
 * the C-level counterpart of lanczosplusplus_amd/comm.py (torch.distributed) so that the C++ host shim
 * (lanczosplusplus_amd/host/lanczos.cpp) can run the row-partitioned path of SURVEY 8(e) without an interpreter.
 *
 *   lpp_rccl_unique_id      rank 0 creates the 128-byte id; the host carries it to the other ranks (file, MPI, socket...)
 *   lpp_rccl_comm_create    allocates the exchange buffers of lpp_comm on `device` and wires the callbacks:
 *       allgather_begin  ncclAllGather of the rank's slice on a side stream (after an event of the compute stream), returns at once
 *       allgather_end    the compute stream waits for the gather's event
 *       allreduce_sum    ncclAllReduce (sum, in place) of red_buf[offset .. offset+count) on the compute stream
 *       exchange_*       the transposition exchange: grouped ncclSend / ncclRecv of nranks equal chunks (an all-to-all)
 *   The all-gather, exchange 0 and exchange 1 each have their own (ready, done) event pair between the two streams.
 *   `stream` is the hipStream_t the engine runs on (pass the same pointer as lpp_config.stream).
 * All functions return an lpp_status; lpp_last_error() of liblpp_engine.so is NOT shared: use lpp_rccl_last_error().
 */
#ifndef LPP_COMM_RCCL_H
#define LPP_COMM_RCCL_H

#include "lpp_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LPP_RCCL_ID_BYTES 128

typedef struct lpp_rccl_comm lpp_rccl_comm;

const char* lpp_rccl_last_error(void);
lpp_status lpp_rccl_unique_id(void* id128);
/* xchg_chunk > 0 selects the transposition exchange (buffers of nranks*xchg_chunk elements), 0 the all-gather
 * (send: shard_stride, gather: nranks*shard_stride elements).  Elements are f64 or (re,im) pairs (is_complex). */
lpp_status lpp_rccl_comm_create(lpp_rccl_comm** out, int32_t rank, int32_t nranks, const void* id128, int32_t device, void* stream,
                                int64_t shard_stride, int32_t max_steps, int32_t is_complex, int64_t xchg_chunk);
/* the communicator in the form the engine takes (valid until lpp_rccl_comm_destroy) */
const lpp_comm* lpp_rccl_comm_get(lpp_rccl_comm* c);
/* Order of destruction: lpp_engine_sync (or lpp_engine_destroy) FIRST, then this.  The communicator never touches the engine's
 * stream here (an engine that owns its stream destroys it): it drains its own side stream and events and frees its buffers, so
 * nothing of the engine may still be queued against them. */
lpp_status lpp_rccl_comm_destroy(lpp_rccl_comm* c);
/* runs every callback once on patterned buffers and checks the results (at nranks == 1 the collectives are identities,
 * which still exercises stream ordering, buffer sizes and the group call): the unit test of the C-level communicator */
lpp_status lpp_rccl_comm_selftest(lpp_rccl_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* LPP_COMM_RCCL_H */
