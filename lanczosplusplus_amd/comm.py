"""torch.distributed communicator for the 1-D row-partitioned multi-GPU path.

One process per GPU.  The exchange buffers are torch tensors (PyTorch owns device memory and
the process group); the engine gets raw pointers plus three callbacks:
  allgather_begin : all_gather_into_tensor(gath, send, async_op=True)  -- returns at once so the
                    local-column SpMV overlaps the transfer
  allgather_end   : work.wait()  -- the engine's stream waits for the gather
  allreduce_sum   : all_reduce on a slice of the scalar buffer (a_j, b_j^2, reortho coefficients)
Backend "nccl" is RCCL over xGMI on MI355X; "gloo" (CPU or GPU tensors) is used by the tests.
The engine must run on the torch stream that is current when the callbacks fire: create the
engine with stream=comm.stream_handle and call it inside `with comm.stream_context():`.

Stream order, per collective (DESIGN.md section 7 has the table for all three communicators): the input is produced on the
engine's stream S; torch.distributed orders its own stream behind an event of the CURRENT stream (= S inside
stream_context()) when the collective is issued and makes the current stream wait for the collective's end event in
work.wait() (nccl: ProcessGroupNCCL's syncStream / WorkNCCL::synchronize; gloo with device tensors: initializeStreamsEvents /
AsyncWork::synchronize), so the consumer kernels, enqueued on S behind the callback, see the result.

LPP_COMM_RECORD=1 (the multi-rank tests set it) records every all-reduce WITHOUT disturbing that order: a copy of the slot on
S in front of the collective (this rank's partial) and one behind it (exactly what the next kernel on S will read).
verify_record() then checks on every rank that each recorded result is the sum of the ranks' partials and bitwise the same
on all ranks -- a wrong energy on several ranks leaves a record of which call, which rank and what it saw.
LPP_GLOO_STREAM_ORDERED=1 makes the gloo emulation hand device tensors to gloo (stream-ordered, like nccl) instead of
reducing / gathering host copies synchronously.
"""
import contextlib
import ctypes as C
import os
import sys
import traceback

import torch
import torch.distributed as dist

from ._capi import CB_REDUCE, CB_VOID, CB_XCHG, Comm


class TorchDistComm:
    def __init__(self, shard_stride, max_steps, is_complex=False, device=None, group=None, xchg_chunk=0):
        """xchg_chunk > 0 selects the transposition exchange (Hubbard device assembly only): two all-to-alls of
        nranks chunks of xchg_chunk = ceil(N_down/P)*ceil(N_up/P) elements replace the all-gather."""
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.device = torch.device(device if device is not None else "cpu")
        ncomp = 2 if is_complex else 1
        self.shard_stride = int(shard_stride)
        self.xchg_chunk = int(xchg_chunk)
        if self.xchg_chunk > 0:
            n = self.nranks * self.xchg_chunk * ncomp
            self.send = torch.zeros(n, dtype=torch.float64, device=self.device)
            self.gath = torch.zeros(n, dtype=torch.float64, device=self.device)
            self.send2 = torch.zeros(n, dtype=torch.float64, device=self.device)
            self.recv2 = torch.zeros(n, dtype=torch.float64, device=self.device)
        else:
            self.send = torch.zeros(self.shard_stride * ncomp, dtype=torch.float64, device=self.device)
            self.gath = torch.zeros(self.nranks * self.shard_stride * ncomp, dtype=torch.float64, device=self.device)
            self.send2 = self.recv2 = None
        self.red = torch.zeros(6 * (max_steps + 2) + 8, dtype=torch.float64, device=self.device)
        self._work = None
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.calls = {"allgather": 0, "allreduce": 0, "exchange": 0}
        self._xwork = None
        self._xstage = None
        # gloo with device buffers (tests, rehearsals with ranks sharing one GPU): host-staged and synchronous by default
        self._host_staged = self.device.type == "cuda" and dist.get_backend(self.group) != "nccl" and os.environ.get("LPP_GLOO_STREAM_ORDERED", "0") == "0"
        # recorder (module docstring): kRecWidth leading doubles of every all-reduce, before and after, copied on the engine's stream
        self._rec_cap = int(os.environ.get("LPP_COMM_RECORD_CALLS", "16384")) if os.environ.get("LPP_COMM_RECORD", "0") != "0" else 0
        self._rec_meta = []
        if self._rec_cap:
            self._rec_in = torch.zeros((self._rec_cap, self.kRecWidth), dtype=torch.float64, device=self.device)
            self._rec_out = torch.zeros((self._rec_cap, self.kRecWidth), dtype=torch.float64, device=self.device)

        def _begin(_ctx):
            try:
                self.calls["allgather"] += 1
                if self._host_staged:
                    # gloo with device buffers (tests: ranks sharing one GPU): gather host copies, see _reduce
                    self._hsend = self.send.cpu()  # waits for the current stream
                    self._hgath = [torch.empty_like(self._hsend) for _ in range(self.nranks)]
                    self._work = dist.all_gather(self._hgath, self._hsend, group=self.group, async_op=True)
                    return 0
                self._hgath = None
                try:
                    self._work = dist.all_gather_into_tensor(self.gath, self.send, group=self.group, async_op=True)
                except (RuntimeError, NotImplementedError):
                    views = list(self.gath.chunk(self.nranks))
                    self._work = dist.all_gather(views, self.send, group=self.group, async_op=True)
                return 0
            except Exception:  # never let an exception cross the C boundary
                traceback.print_exc(file=sys.stderr)
                return 1

        def _end(_ctx):
            try:
                if self._work is not None:
                    self._work.wait()
                    self._work = None
                if getattr(self, "_hgath", None) is not None:
                    for q, t in enumerate(self.gath.chunk(self.nranks)):
                        t.copy_(self._hgath[q])
                    self._hgath = None
                return 0
            except Exception:
                traceback.print_exc(file=sys.stderr)
                return 1

        def _reduce(_ctx, offset, count):
            try:
                self.calls["allreduce"] += 1
                view = self.red[offset:offset + count]
                k = len(self._rec_meta)
                rec = k < self._rec_cap
                w = min(count, self.kRecWidth)
                if rec:
                    self._rec_in[k, :w].copy_(view[:w])  # on the engine's stream, no host synchronisation
                if self._host_staged:
                    # gloo (tests, rehearsals with ranks sharing one GPU): reduce a host copy, synchronously on both sides
                    host = view.cpu()  # waits for the current stream
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(host)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                if rec:
                    self._rec_out[k, :w].copy_(view[:w])  # what the next kernel on this stream will read
                    self._rec_meta.append((int(offset), int(count)))
                return 0
            except Exception:
                traceback.print_exc(file=sys.stderr)
                return 1

        def _xbegin(_ctx, which):
            """all-to-all of nranks equal chunks: chunk p of the source goes to rank p, chunk q of the destination
            comes from rank q.  nccl: all_to_all_single; gloo (tests): point-to-point with host staging."""
            try:
                self.calls["exchange"] += 1
                src, dst = (self.send, self.gath) if which == 0 else (self.send2, self.recv2)
                if dist.get_backend(self.group) == "nccl":
                    self._xwork = dist.all_to_all_single(dst, src, group=self.group, async_op=True)
                    self._xstage = None
                else:
                    sc, dc = src.chunk(self.nranks), dst.chunk(self.nranks)
                    dc[self.rank].copy_(sc[self.rank])
                    outs = {p: sc[p].cpu() for p in range(self.nranks) if p != self.rank}
                    ins = {p: torch.empty_like(outs[p]) for p in outs}
                    reqs = []
                    for p in outs:
                        reqs.append(dist.isend(outs[p], dst=p, group=self.group))
                        reqs.append(dist.irecv(ins[p], src=p, group=self.group))
                    self._xwork = reqs
                    self._xstage = (ins, dc, outs)
                return 0
            except Exception:
                traceback.print_exc(file=sys.stderr)
                return 1

        def _xend(_ctx, which):
            try:
                if self._xstage is None:
                    if self._xwork is not None:
                        self._xwork.wait()
                else:
                    for r in self._xwork:
                        r.wait()
                    ins, dc, _ = self._xstage
                    for p, t in ins.items():
                        dc[p].copy_(t)
                self._xwork = None
                self._xstage = None
                return 0
            except Exception:
                traceback.print_exc(file=sys.stderr)
                return 1

        # keep references: ctypes callbacks must outlive the engine
        self._cb = (CB_VOID(_begin), CB_VOID(_end), CB_REDUCE(_reduce), CB_XCHG(_xbegin), CB_XCHG(_xend))
        s = Comm()
        s.rank, s.nranks, s.ctx = self.rank, self.nranks, None
        s.send_buf = self.send.data_ptr()
        s.gath_buf = self.gath.data_ptr()
        s.red_buf = self.red.data_ptr()
        s.shard_stride = self.shard_stride
        s.red_len = self.red.numel()
        s.allgather_begin, s.allgather_end, s.allreduce_sum = self._cb[:3]
        if self.xchg_chunk > 0:
            s.send2_buf = self.send2.data_ptr()
            s.recv2_buf = self.recv2.data_ptr()
            s.xchg_chunk = self.xchg_chunk
            s.exchange_begin, s.exchange_end = self._cb[3], self._cb[4]
        self.struct = s

    kRecWidth = 4

    def verify_record(self, rtol=1e-12):
        """Collective (every rank calls it, outside a run): checks the recorded all-reduces.  Returns a list of findings, empty
        when every recorded result is the sum of the ranks' partials (to rounding: the backend's summation order is its own)
        and bitwise identical on all ranks.  A finding names the call, its slot in the scalar buffer, the rank that saw a wrong
        value, that value, the expected sum and every rank's partial."""
        if not self._rec_cap:
            return []
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        n = len(self._rec_meta)
        mine = {"meta": list(self._rec_meta), "in": self._rec_in[:n].cpu().numpy(), "out": self._rec_out[:n].cpu().numpy()}
        everyone = [None] * self.nranks
        dist.all_gather_object(everyone, mine, group=self.group)
        findings = []
        metas = [tuple(r["meta"]) for r in everyone]
        if any(m != metas[0] for m in metas):
            findings.append({"what": "the ranks issued different all-reduce sequences", "calls": [len(m) for m in metas]})
            n = min(len(m) for m in metas)
        import numpy as np
        for k in range(n):
            off, cnt = metas[0][k]
            w = min(cnt, self.kRecWidth)
            parts = np.stack([r["in"][k, :w] for r in everyone])
            want = parts.sum(axis=0)
            scale = np.abs(parts).sum(axis=0) + 1e-300
            for q, r in enumerate(everyone):
                got = r["out"][k, :w]
                bad = ~(np.abs(got - want) <= rtol * scale)  # catches NaN too
                differs = got.tobytes() != everyone[0]["out"][k, :w].tobytes()
                if bad.any() or differs:
                    findings.append({"what": "wrong sum" if bad.any() else "ranks disagree bitwise", "call": k, "offset": off, "count": cnt, "rank": q,
                                     "saw": got.tolist(), "expected": want.tolist(), "partials": parts.tolist()})
        self._rec_meta = []
        return findings

    @property
    def stream_handle(self):
        return None if self.stream is None else C.c_void_p(self.stream.cuda_stream)

    def stream_context(self):
        if self.stream is None:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.stream)


class RcclComm:
    """The C-level communicator of include/lpp_comm_rccl.h (liblpp_comm_rccl.so) held from Python: the collectives of a Lanczos
    step are issued from C on HIP streams -- no interpreter, no ctypes trampoline and no torch dispatch between the kernels of
    a step (TorchDistComm pays those seven times per step).  The 128-byte RCCL id is carried to the other ranks by the caller
    (bench.py broadcasts it with torch.distributed).  One GPU per rank (RCCL cannot put two ranks on one device)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import os
            here = os.path.dirname(os.path.abspath(__file__))
            L = C.CDLL(os.path.join(here, "csrc", "liblpp_comm_rccl.so"))
            L.lpp_rccl_last_error.restype = C.c_char_p
            L.lpp_rccl_unique_id.argtypes = [C.c_void_p]
            L.lpp_rccl_comm_get.restype = C.POINTER(Comm)
            L.lpp_rccl_comm_get.argtypes = [C.c_void_p]
            L.lpp_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                               C.c_int32, C.c_int32, C.c_int64]
            L.lpp_rccl_comm_destroy.argtypes = [C.c_void_p]
            L.lpp_rccl_comm_selftest.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    @classmethod
    def unique_id(cls):
        buf = C.create_string_buffer(128)
        if cls.lib().lpp_rccl_unique_id(buf) != 0:
            raise RuntimeError(cls.lib().lpp_rccl_last_error().decode())
        return buf.raw

    def __init__(self, rank, nranks, ident, device, stream_ptr, shard_stride, max_steps, is_complex=False, xchg_chunk=0):
        L = self.lib()
        self._h = C.c_void_p()
        self.rank, self.nranks = int(rank), int(nranks)
        self.shard_stride, self.xchg_chunk = int(shard_stride), int(xchg_chunk)
        self._ident = C.create_string_buffer(bytes(ident), 128)
        rc = L.lpp_rccl_comm_create(C.byref(self._h), self.rank, self.nranks, self._ident, int(device), stream_ptr, self.shard_stride,
                                    int(max_steps), int(bool(is_complex)), self.xchg_chunk)
        if rc != 0:
            raise RuntimeError("lpp_rccl_comm_create: " + L.lpp_rccl_last_error().decode())
        self.struct = L.lpp_rccl_comm_get(self._h).contents
        ncomp = 2 if is_complex else 1
        n = (self.nranks * self.xchg_chunk if self.xchg_chunk > 0 else self.shard_stride) * ncomp
        g = (self.nranks * self.xchg_chunk if self.xchg_chunk > 0 else self.nranks * self.shard_stride) * ncomp
        self.buffer_bytes = 8 * (n + g + (2 * n if self.xchg_chunk > 0 else 0))
        self.calls = None  # issued from C: not counted

    def selftest(self):
        """every callback once on patterned buffers, results checked on every rank (collective: all ranks must call it)"""
        if self.lib().lpp_rccl_comm_selftest(self._h) != 0:
            raise RuntimeError("lpp_rccl_comm_selftest: " + self.lib().lpp_rccl_last_error().decode())

    def stream_context(self):
        return contextlib.nullcontext()

    def close(self):
        if self._h:
            self.lib().lpp_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()


class ThreadGroup:
    """N ranks as THREADS of one process that share one GPU (a rehearsal of a rank count a one-GPU box cannot host as processes: at most
    6 processes may use its card).  Every rank runs its own engine on its own stream; the collectives are device-to-device copies and host
    sums between the ranks' buffers, ordered by a barrier.  Same callbacks, buffers and wire format as the other communicators -- the
    engine cannot tell the difference -- so the kernels run with the geometry of the real rank count (shards, chunks, up ranges)."""

    def __init__(self, nranks, device):
        import threading
        self.nranks = int(nranks)
        self.device = torch.device(device)
        self.barrier = threading.Barrier(self.nranks)
        self.comms = [None] * self.nranks
        self.parts = [None] * self.nranks

    def wait(self):
        self.barrier.wait(timeout=600)


class ThreadComm:
    def __init__(self, group, rank, shard_stride, max_steps, is_complex=False, xchg_chunk=0):
        self.group, self.rank, self.nranks = group, int(rank), group.nranks
        self.device = group.device
        ncomp = 2 if is_complex else 1
        self.shard_stride, self.xchg_chunk = int(shard_stride), int(xchg_chunk)
        z = lambda n: torch.zeros(n, dtype=torch.float64, device=self.device)
        if self.xchg_chunk > 0:
            n = self.nranks * self.xchg_chunk * ncomp
            self.send, self.gath, self.send2, self.recv2 = z(n), z(n), z(n), z(n)
        else:
            self.send, self.gath = z(self.shard_stride * ncomp), z(self.nranks * self.shard_stride * ncomp)
            self.send2 = self.recv2 = None
        self.red = z(6 * (max_steps + 2) + 8)
        self.stream = torch.cuda.Stream(device=self.device)
        self.calls = {"allgather": 0, "allreduce": 0, "exchange": 0}
        self.sums = []  # every all-reduce as (offset, this rank's partial, the sum it got back)
        group.comms[self.rank] = self

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a)
                except Exception:
                    traceback.print_exc(file=sys.stderr)
                    try:
                        self.group.barrier.abort()  # never leave the other ranks inside a collective
                    except Exception:
                        pass
                    return 1
            return wrapped

        def _begin(_ctx):
            self.calls["allgather"] += 1
            self.stream.synchronize()  # this rank's slice is complete
            self.group.wait()
            with torch.cuda.stream(self.stream):
                for q, t in enumerate(self.gath.chunk(self.nranks)):
                    t.copy_(self.group.comms[q].send)
            return 0

        def _end(_ctx):
            self.stream.synchronize()
            self.group.wait()  # nobody overwrites a slice a peer is still copying
            return 0

        def _reduce(_ctx, offset, count):
            self.calls["allreduce"] += 1
            view = self.red[offset:offset + count]
            with torch.cuda.stream(self.stream):
                mine = view.cpu()  # waits for this rank's stream
            self.group.parts[self.rank] = mine
            self.group.wait()
            total = self.group.parts[0].clone()
            for q in range(1, self.nranks):  # the same order on every rank: bitwise identical sums, identical decisions
                total += self.group.parts[q]
            self.group.wait()  # everyone has read every partial
            with torch.cuda.stream(self.stream):
                view.copy_(total)
            self.sums.append((int(offset), mine.clone(), total))
            return 0

        def _xbegin(_ctx, which):
            self.calls["exchange"] += 1
            self.stream.synchronize()  # this rank's chunks are packed
            self.group.wait()
            with torch.cuda.stream(self.stream):
                dst = self.gath if which == 0 else self.recv2
                for q, t in enumerate(dst.chunk(self.nranks)):
                    peer = self.group.comms[q]
                    t.copy_((peer.send if which == 0 else peer.send2).chunk(self.nranks)[self.rank])
            return 0

        def _xend(_ctx, which):
            self.stream.synchronize()
            self.group.wait()
            return 0

        self._cb = (CB_VOID(guard(_begin)), CB_VOID(guard(_end)), CB_REDUCE(guard(_reduce)), CB_XCHG(guard(_xbegin)), CB_XCHG(guard(_xend)))
        s = Comm()
        s.rank, s.nranks, s.ctx = self.rank, self.nranks, None
        s.send_buf, s.gath_buf, s.red_buf = self.send.data_ptr(), self.gath.data_ptr(), self.red.data_ptr()
        s.shard_stride, s.red_len = self.shard_stride, self.red.numel()
        s.allgather_begin, s.allgather_end, s.allreduce_sum = self._cb[:3]
        if self.xchg_chunk > 0:
            s.send2_buf, s.recv2_buf, s.xchg_chunk = self.send2.data_ptr(), self.recv2.data_ptr(), self.xchg_chunk
            s.exchange_begin, s.exchange_end = self._cb[3], self._cb[4]
        self.struct = s

    @property
    def stream_handle(self):
        return C.c_void_p(self.stream.cuda_stream)

    def stream_context(self):
        return torch.cuda.stream(self.stream)
