"""start the recurrence from a converged eigenvector (lucky breakdown: b_0 ~ 0): chained product-basis step vs general layout"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import chain, square
from lanczosplusplus_amd import LanczosEngine
L, nup, ndown = 12, 6, 5
hop, U = chain(L, -1.0, True), np.full(L, 4.0)
for layout in ("1", "0"):
    os.environ["LPP_PRODUCT_LAYOUT"] = layout
    with LanczosEngine(max_steps=300, eps=1e-13) as e:
        e.assemble_hubbard(L, nup, ndown, hop, U)
        eg, zg, st = e.lanczos(1, want_vectors=True)
        for sv in (0, 1):
            with LanczosEngine(max_steps=50, save_vectors=sv) as e2:
                e2.assemble_hubbard(L, nup, ndown, hop, U)
                a, b, st2 = e2.decomposition(zg[0])
                print("layout", layout, "kernel", e2.layout()["kernel"], "save_vectors", sv, "E0", eg[0], "steps", len(a), "a0", a[0], "b0", b[0], "finite", bool(np.all(np.isfinite(a)) and np.all(np.isfinite(b))))
