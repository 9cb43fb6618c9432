"""Free-fermion check of the matrix-free engine at sizes no CPU can follow: U=0 on a periodic lx x ly lattice, E0 must equal
the sum of the lowest single-particle levels of both species.
  python scripts/big_free_fermions.py            3x6, 9 up 9 down: 2.36e9 states
  python scripts/big_free_fermions.py 4 5 8 7    4x5, 8 up 7 down: 9.77e9 states (78 GB per vector)"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import square_lattice
from lanczosplusplus_amd import LanczosEngine
lx, ly, nu, nd = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (3, 6, 9, 9)
L = lx * ly
hop = square_lattice(lx, ly, -1.0)
lev = np.sort(np.linalg.eigvalsh(hop))
exact = lev[:nu].sum() + lev[:nd].sum()
t0 = time.time()
with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
    e.setup_hubbard_onthefly(L, nu, nd, hop, np.zeros(L))
    eg, _, st = e.lanczos(1, want_vectors=False)
print("rows", st["nrows"], "steps", st["steps"], "E0 %.12f" % eg[0], "exact %.12f" % exact, "rel diff %.2e" % ((eg[0] - exact) / abs(exact)),
      "time %.1f s" % (time.time() - t0), flush=True)
