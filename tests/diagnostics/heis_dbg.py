import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle
from helpers import chain
from lanczosplusplus_amd import LanczosEngine
L, twiceS, m = 6, 3, 9
rng = np.random.default_rng(100 * L + twiceS)
jpm, jzz = chain(L, 1.0, False), chain(L, 0.7 + 0.1 * twiceS, False)
jpm[0, 1] = jpm[1, 0] = 1.37
field = rng.uniform(-0.3, 0.3, L)
A = oracle.heis_csr(L, twiceS, m, jpm, jzz, field=field)
with LanczosEngine() as e:
    e.assemble_heisenberg(L, m, jpm, jzz, field, twiceS=twiceS)
    rp, ci, va = e.get_csr()
bad = np.nonzero(va.view(np.uint64) != A.values.view(np.uint64))[0]
print("differing", len(bad), "of", len(va))
rows = np.searchsorted(rp, bad, side="right") - 1
for k, r in list(zip(bad, rows))[:20]:
    print(r, ci[k], repr(va[k]), repr(A.values[k]), "diag" if ci[k] == r else "off")
print("jzz01", repr(jzz[0, 1]))
