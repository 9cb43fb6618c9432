"""2.36e9-state check of the matrix-free engine: free fermions (U=0) on the periodic 3x6 lattice, 9 up 9 down.
E0 must equal twice the sum of the nine lowest single-particle levels."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import square_lattice
from lanczosplusplus_amd import LanczosEngine
L = 18
hop = square_lattice(3, 6, -1.0)
lev = np.sort(np.linalg.eigvalsh(hop))
exact = 2 * lev[:9].sum()
t0 = time.time()
with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
    e.setup_hubbard_onthefly(L, 9, 9, hop, np.zeros(L))
    eg, _, st = e.lanczos(1, want_vectors=False)
print("rows", st["nrows"], "steps", st["steps"], "E0", eg[0], "exact", exact, "diff", eg[0] - exact, "time", time.time() - t0)
