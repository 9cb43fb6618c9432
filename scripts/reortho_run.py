"""Full re-orthogonalisation run for profiling the blocked Gram-Schmidt kernels: Heisenberg chain L=28, Sz=0
(4.0e7 states), 64 Lanczos steps with CGS2 against the on-device Krylov basis."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import chain
from lanczosplusplus_amd import LanczosEngine
L = 28
with LanczosEngine(max_steps=64, eps=0.0, reortho=True, save_vectors=1) as e:
    e.assemble_heisenberg(L, 14, chain(L, 1.0), chain(L, 1.0))
    t0 = time.time()
    a, b, st = e.decomposition()
    dt = time.time() - t0
print("rows", st["nrows"], "steps", st["steps"], "seconds %.3f" % dt, "a0", a[0], "b_last", b[-1])
