"""Step time of one workload on the engine library LPP_ENGINE_LIB names (default: the in-tree one), without bench.py's
coefficient gate: for timing-only builds whose results are wrong by construction.  Prints ms per chained Lanczos step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from bench import WORKLOADS
from lanczosplusplus_amd.engine import LanczosEngine

name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half_filling_pbc_U4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kind, p = WORKLOADS[name]
L = p["L"]
with LanczosEngine(dtype="c128" if np.iscomplexobj(p["hop"]()) else "f64") as e:
    if os.environ.get("OTF", "0") == "1":
        e.setup_hubbard_onthefly(L, p["nup"], p["ndown"], p["hop"](), np.full(L, p["U"]), np.zeros(L))
    else:
        e.assemble_hubbard(L, p["nup"], p["ndown"], p["hop"](), np.full(L, p["U"]), np.zeros(L))
    e.begin()
    e.step(5)
    e.sync()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.step(steps)
        e.sync()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    a, b = e.coeffs()
    print("%s lib=%s: %.3f ms per step (best of 3 x %d); a[3]=%.6f" % (name, os.path.basename(os.environ.get("LPP_ENGINE_LIB", "in-tree")), best, steps, a[3]))
