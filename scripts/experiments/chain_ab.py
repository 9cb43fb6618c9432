"""chained vs three-kernel scale-free step on the same problem: coefficient-by-coefficient difference"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from helpers import chain, square
    from lanczosplusplus_amd import LanczosEngine, tridiag_lowest
    L = 12
    hop, U = chain(L, -1.0), np.full(L, 4.0)
    with LanczosEngine(max_steps=300, eps=1e-12, save_vectors=0) as e:
        e.assemble_hubbard(L, 6, 6, hop, U)
        a, b, st = e.decomposition()
    print(json.dumps({"a": list(a), "b": list(b)}))
else:
    out = {}
    for v in ("1", "0"):
        env = dict(os.environ, LPP_PB_CHAIN=v)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        out[v] = json.loads(r.stdout.strip().splitlines()[-1])
    a1, a0 = np.array(out["1"]["a"]), np.array(out["0"]["a"])
    b1, b0 = np.array(out["1"]["b"]), np.array(out["0"]["b"])
    n = min(len(a1), len(a0))
    print("steps", len(a1), len(a0))
    for k in range(0, n, max(1, n // 40)):
        print(k, a1[k], a0[k], abs(a1[k] - a0[k]), b1[k], b0[k], abs(b1[k] - b0[k]))
