#!/usr/bin/env python3
"""Generates tests/golden/c2_hubbard4x4_U4.json: the CPU-oracle Lanczos run of BASELINE config 2.

2-D Hubbard 4x4, periodic, t = 1 (hoppings -1), U = 4, 8 up / 8 down: 165,636,900 states, 5.8e9 non-zeros.  The
reference's stored path cannot hold it (int row pointers); its CPU path at this size is the on-the-fly
product (SolverOptions=InternalProductOnTheFly -> HubbardHelper::matrixVectorProduct, HubbardHelper.h:105-134).
The oracle restates that product twice: literally (lppo_hubbard_otf_mvp) and tabulated (lppo_hubbard_otf_apply, the
same elements in the same summation order, bit-identical: tests/test_oracle_pins.py and the sample check below).
The Lanczos loop is the oracle's (lppo_lanczos_decomposition's recurrence), start vector = the built-in splitmix64
stream with seed 1234 that the GPU engine generates for init == NULL, eps = 1e-12, minSteps = 4, maxSteps = 300.

Run in the build container (about 10 minutes on 8 cores, 6 GB):   python tests/golden/make_c2_hubbard4x4_U4.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from helpers import square  # noqa: E402


def main():
    L, nup, ndown, Uval = 16, 8, 8, 4.0
    threads = int(os.environ.get("LPP_CPU_THREADS", "8"))
    hop, U, V = square(4, 4, -1.0, pbc=True), np.full(L, Uval), np.zeros(L)
    H = oracle.HubbardOtf(L, nup, ndown, hop, U, V, nthreads=threads)
    n = H.nrows
    assert n == 165636900
    init = oracle.fill_random(n, 1234)
    # the tabulated product against the literal restatement on three row windows of the full-size problem
    y = oracle.fill_random(n, 99)
    windows = [(0, 150_000), (n // 2 - 7, n // 2 + 150_000), (n - 150_000, n)]
    for (r0, r1) in windows:
        xa, xb = np.zeros(n), np.zeros(n)
        oracle.hubbard_otf_mvp(L, nup, ndown, hop, U, V, xa, y, r0, r1, threads)
        H.apply(xb, y, r0, r1)
        assert np.array_equal(xa[r0:r1], xb[r0:r1]) and not np.any(xb[:r0]) and not np.any(xb[r1:])
    del y, xa, xb
    t0 = time.time()
    steps, a, b, hist = H.lanczos(init, max_steps=300, min_steps=4, eps=1e-12)
    dt = time.time() - t0
    out = {
        "what": "oracle (CPU) on-the-fly Lanczos of the 4x4 periodic Hubbard cluster, t=1, U=4, 8 up 8 down",
        "generator": "tests/golden/make_c2_hubbard4x4_U4.py",
        "L": L, "nup": nup, "ndown": ndown, "U": Uval, "lattice": "square 4x4 periodic, hopping -1",
        "rows": int(n), "seed": 1234, "eps": 1e-12, "min_steps": 4, "max_steps": 300,
        "steps": int(steps), "e0": float(hist[-1]),
        "a": [float(v) for v in a], "b": [float(v) for v in b], "e0_history": [float(v) for v in hist],
        "literal_sample_rows": [list(w) for w in windows],
        "cpu_seconds": round(dt, 1), "cpu_threads": threads,
    }
    path = os.path.join(ROOT, "tests", "golden", "c2_hubbard4x4_U4.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s: %d steps, E0 = %.12f, %.0f s" % (path, steps, hist[-1], dt))


if __name__ == "__main__":
    main()
