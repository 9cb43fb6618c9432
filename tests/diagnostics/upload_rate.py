"""PCIe-inclusive cost of the host-CSR boundary: lpp_engine_set_csr of a host-assembled Hubbard chain L=14 (1.18e7 rows,
1.77e8 non-zeros, 2.2 GB of CSR) -- upload + layout conversion -- then the solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from lanczosplusplus_amd import LanczosEngine
from math import comb

if len(sys.argv) > 1 and sys.argv[1] == "4x4":
    # 4x4 cluster, 6 up / 6 down: 6.4e7 states, ~2e9 non-zeros, ~25 GB of host CSR
    from bench import square_lattice
    L, nu, nd = 16, 6, 6
    hop = square_lattice(4, 4, -1.0)
else:
    L, nu, nd = 14, 7, 7
    hop = np.zeros((L, L))
    for i in range(L - 1):
        hop[i, i + 1] = hop[i + 1, i] = -1.0
t0 = time.time()
A = oracle.hubbard_csr(L, nu, nd, hop, np.full(L, 4.0))
t_asm = time.time() - t0
nbytes = A.rowptr.nbytes + A.colind.nbytes + A.values.nbytes
with LanczosEngine(max_steps=300) as e:
    e.set_row_block(comb(L, nu))
    t0 = time.time()
    e.set_csr(A.rowptr, A.colind, A.values)
    e.sync()
    t_set = time.time() - t0
    t0 = time.time()
    eg, _, st = e.lanczos(1, want_vectors=False)
    t_solve = time.time() - t0
    lay = e.layout()
    ms = e.bench_spmv(3, 20)
print("rows %d nnz %d csr %.2f GB | host assembly (oracle, 1 thread) %.1f s | set_csr (PCIe upload + layout) %.2f s = %.1f GB/s | "
      "solve %d steps %.3f s E0=%.10f | SpMV %.3f ms | layout %s" % (A.nrows, A.nnz, nbytes / 1e9, t_asm, t_set, nbytes / 1e9 / t_set, st["steps"], t_solve, eg[0], ms, lay))
