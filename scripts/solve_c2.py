"""End-to-end convergence-checked solve on config 2 (both engines): steps, wall time, SpMV share."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import square_lattice
from lanczosplusplus_amd import LanczosEngine
L = 16
hop, U = square_lattice(4, 4, -1.0), np.full(L, 4.0)
for engine in ("stored", "onthefly"):
    with LanczosEngine(max_steps=300, save_vectors=0, time_kernels=True) as e:
        t0 = time.time()
        (e.assemble_hubbard if engine == "stored" else e.setup_hubbard_onthefly)(L, 8, 8, hop, U)
        e.sync()
        t1 = time.time()
        eg, _, st = e.lanczos(1, want_vectors=False)
        t2 = time.time()
        print(engine, "E0=%.12f" % eg[0], "steps", st["steps"], "enqueued", st["steps_enqueued"], "setup %.2fs" % (t1 - t0),
              "solve %.3fs" % (t2 - t1), "spmv %.3fs" % (st["spmv_ms_total"] / 1e3), "it/s %.1f" % (st["steps_enqueued"] / (t2 - t1)))
        t3 = time.time()
        eg, zg, st = e.lanczos(1, want_vectors=True)
        print(engine, "with ground-state vector (two-pass):", "%.3fs" % (time.time() - t3), "norm", np.linalg.norm(zg[0]))
