"""Complex hoppings beyond one LDS window at full size (4x4 lattice, Peierls phases, 8 up 7 down: 1.47e8 complex states): the pieces form of the
product-basis layout against the general layout -- resident bytes, time per step, exact free-fermion energy.  Run on the GPU box."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from lanczosplusplus_amd import LanczosEngine
from test_gpu_fullsize import square, _exact
L = 16
hop = square(4, 4, -1.0, pbc=True) * np.where(np.triu(np.ones((L, L)), 1) > 0, np.exp(0.2j), np.exp(-0.2j))
nu, nd = 8, 7
exact = 0.5 * (_exact(hop, nu) + _exact(hop, nd)) if False else None
w = np.linalg.eigvalsh(hop)
exact = w[:nu].sum() + w[:nd].sum()
for env in (({},) if "--product-only" in sys.argv else ({}, {"LPP_PB_COMPLEX": "0"})):
    os.environ.update(env)
    t0 = time.time()
    with LanczosEngine(dtype="c128", max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, nu, nd, hop, np.zeros(L))
        lay = e.layout()
        t1 = time.time()
        eg, _, st = e.lanczos(1, want_vectors=False)
        t2 = time.time()
        print(env, "kernel", lay["kernel"], "pieces", lay["pieces"], "resident GB", lay["resident_bytes"] / 1e9, "assemble s", t1 - t0, "steps", st["steps"], "solve s", t2 - t1,
              "ms/step", 1e3 * (t2 - t1) / st["steps"], "E", eg[0], "exact", exact, "rel", abs(eg[0] - exact) / abs(exact), flush=True)
