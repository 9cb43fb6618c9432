"""What ONE rank of a P-rank run computes per Lanczos step, timed alone on one GPU: a communicator whose collectives do nothing
(the exchange buffers keep whatever they hold, so the numbers of the run mean nothing -- the kernels, their sizes and their order
are those of rank 0 of a real P-rank run).  Gives the compute term of the scaling model in DESIGN.md section 7; the transfer term
needs the real node.      python tests/diagnostics/rank_share_timing.py [P ...]"""
import ctypes as C
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from bench import WORKLOADS
from lanczosplusplus_amd import LanczosEngine
from lanczosplusplus_amd._capi import CB_REDUCE, CB_VOID, CB_XCHG, Comm, lib
from math import comb


class NullComm:
    def __init__(self, rank, nranks, stride, max_steps, chunk, dev):
        n = nranks * chunk if chunk > 0 else stride
        g = nranks * chunk if chunk > 0 else nranks * stride
        self.send = torch.zeros(n, dtype=torch.float64, device=dev)
        self.gath = torch.full((g,), 1e-3, dtype=torch.float64, device=dev)
        self.send2 = torch.zeros(n, dtype=torch.float64, device=dev) if chunk > 0 else None
        self.recv2 = torch.full((g,), 1e-3, dtype=torch.float64, device=dev) if chunk > 0 else None
        self.red = torch.zeros(6 * (max_steps + 2) + 8, dtype=torch.float64, device=dev)
        self._cb = (CB_VOID(lambda c: 0), CB_VOID(lambda c: 0), CB_REDUCE(lambda c, o, k: 0), CB_XCHG(lambda c, w: 0), CB_XCHG(lambda c, w: 0))
        s = Comm()
        s.rank, s.nranks, s.ctx = rank, nranks, None
        s.send_buf, s.gath_buf, s.red_buf = self.send.data_ptr(), self.gath.data_ptr(), self.red.data_ptr()
        s.shard_stride, s.red_len = stride, self.red.numel()
        s.allgather_begin, s.allgather_end, s.allreduce_sum = self._cb[:3]
        if chunk > 0:
            s.send2_buf, s.recv2_buf, s.xchg_chunk = self.send2.data_ptr(), self.recv2.data_ptr(), chunk
            s.exchange_begin, s.exchange_end = self._cb[3], self._cb[4]
        self.struct, self.xchg_chunk = s, chunk


def main():
    # python tests/diagnostics/rank_share_timing.py [--onthefly] [--workload NAME] [P ...]
    onthefly = "--onthefly" in sys.argv
    name = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "hubbard_4x4_half_filling_pbc_U4"
    sys.argv = [a for i, a in enumerate(sys.argv) if a != "--onthefly" and a != "--workload" and (i == 0 or sys.argv[i - 1] != "--workload")]
    model, p = WORKLOADS[name]
    n_up, n_dn = comb(p["L"], p["nup"]), comb(p["L"], p["ndown"])
    hop, U = p["hop"](), np.full(p["L"], p["U"])
    steps, warm = 30, 5
    for P in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        for exchange in (("transpose",) if onthefly else ("transpose", "allgather")):
            per = -(-n_dn // P)
            chunk = lib().lpp_xchg_chunk(n_up, n_dn, P) if exchange == "transpose" else 0
            c = NullComm(0, P, per * n_up, steps + warm + 2, chunk, "cuda")
            with LanczosEngine(max_steps=steps + warm + 2, eps=0.0, save_vectors=0, time_kernels=True) as e:
                t0 = time.time()
                if onthefly:
                    e.setup_hubbard_onthefly(p["L"], p["nup"], p["ndown"], hop, U, comm=c)
                else:
                    e.assemble_hubbard(p["L"], p["nup"], p["ndown"], hop, U, comm=c)
                e.sync()
                t_asm = time.time() - t0
                lay = e.layout(0)["kernel"] if not onthefly or True else 0
                e.begin(None)
                e.step(warm)
                e.sync()
                e.stats()
                w0 = e.stats()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e.step(steps)
                e.sync()
                dt = (time.perf_counter() - t0) / steps
                w1 = e.stats()
            print("P=%d %-9s layout kernel %d: %.3f ms per step on rank 0 alone (product kernels %.3f ms), assembly %.2f s"
                  % (P, exchange, lay, 1e3 * dt, (w1["spmv_ms_total"] - w0["spmv_ms_total"]) / steps, t_asm), flush=True)


if __name__ == "__main__":
    main()
