"""The two species shapes of BASELINE config 5 as literally written (4x5, 10 up 10 down) on sectors that fit one GPU:
(10,2): rows of 184756 positions (the in-block kernel's shape), (2,10): 184756 blocks (the coupling kernel's shape)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from helpers import square
from lanczosplusplus_amd import LanczosEngine

L = 20
hop = square(4, 5, -1.0, pbc=True)
lev = np.sort(np.linalg.eigvalsh(hop))
for nup, ndown in [tuple(int(x) for x in s.split(",")) for s in (sys.argv[1:] or ["10,2", "2,10", "10,3"])]:
    exact = lev[:nup].sum() + lev[:ndown].sum()
    for U in (0.0, 4.0):
        res = {}
        for kron in ("0", "1"):
            if U == 0.0 and kron == "1":
                continue
            os.environ["LPP_ONTHEFLY_KRON"] = kron
            with LanczosEngine(max_steps=400, eps=1e-11, save_vectors=0) as e:
                t0 = time.time()
                e.setup_hubbard_onthefly(L, nup, ndown, hop, np.full(L, U))
                try:
                    lay = e.layout()
                    desc = {k: lay[k] for k in ("kernel", "pieces", "segments", "coupling_parts", "coupling_rounds")}
                except Exception as ex:
                    desc = "no layout (fused kernel)"
                eg, _, st = e.lanczos(1, want_vectors=False)
                res[kron] = eg[0]
                print("(%d,%d) U=%g kron=%s rows=%d: E0=%.12f steps=%d %.1fs %s" % (nup, ndown, U, kron, e.rows(), eg[0], st["steps"], time.time() - t0, desc), flush=True)
        if U == 0.0:
            print("     exact %.12f  diff %.2e" % (exact, abs(res["0"] - exact)))
        else:
            print("     product-basis vs fused kernel: %.2e" % abs(res["0"] - res["1"]))
os.environ.pop("LPP_ONTHEFLY_KRON", None)
