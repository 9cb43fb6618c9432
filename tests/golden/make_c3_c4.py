#!/usr/bin/env python3
"""Generates tests/golden/c3_heisenberg_L28.json and tests/golden/c4_tj_4x5_complex.json: the CPU-oracle Lanczos runs of
BASELINE configs 3 and 4 at full size, in the format of c2_hubbard4x4_U4.json (bench.py prints |E0(GPU) - E0(CPU)| from them).

  C3  Heisenberg S=1/2 open chain, L = 28, Sz = 0, J+- = Jzz = 1: 40,116,600 states, 601,749,000 non-zeros
      (Heisenberg.h:80-114 through the oracle's heis_csr, rank by the combinatorial formula == the reference's O(N) scan).
  C4  t-J 4x5 (5 x 4 sites, periodic), 9 up 9 down, t = 1 (hopping -1), J = 0.4, W = -J/4, complex<double>:
      9,237,800 states (TjMultiOrb.h:100-131 through the oracle's tj_csr).
Stored product (the reference's InternalProductStored path) with the oracle's threaded x += A y, the oracle's Lanczos loop, start
vector = the built-in splitmix64 stream with seed 1234, eps = 1e-12, minSteps = 4, maxSteps = 300.

Run in the build container (a few minutes on 8 cores, ~12 GB):   python tests/golden/make_c3_c4.py
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
from helpers import chain, square  # noqa: E402


def run(name, what, A, complex_init, extra):
    threads = int(os.environ.get("LPP_CPU_THREADS", "8"))
    init = oracle.fill_random(A.nrows, 1234, complex_init)
    t0 = time.time()
    steps, a, b, _, hist = oracle.lanczos_decomposition(A, init, max_steps=300, min_steps=4, eps=1e-12, nthreads=threads)
    dt = time.time() - t0
    out = {"what": what, "generator": "tests/golden/make_c3_c4.py", "rows": int(A.nrows), "nnz": int(A.nnz), "seed": 1234, "eps": 1e-12,
           "min_steps": 4, "max_steps": 300, "steps": int(steps), "e0": float(hist[-1]), "a": [float(v) for v in a], "b": [float(v) for v in b],
           "e0_history": [float(v) for v in hist], "cpu_seconds": round(dt, 1), "cpu_threads": threads}
    out.update(extra)
    path = os.path.join(ROOT, "tests", "golden", name)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s: %d steps, E0 = %.12f, %.0f s" % (path, steps, hist[-1], dt))


def main():
    L = 28
    A = oracle.heis_csr(L, 1, 14, chain(L, 1.0), chain(L, 1.0))
    assert (A.nrows, A.nnz) == (40116600, 601749000)
    run("c3_heisenberg_L28.json", "oracle (CPU) stored-CSR Lanczos of the open S=1/2 Heisenberg chain, L=28, Sz=0, J=1", A, False,
        {"L": L, "szPlusConst": 14, "J": 1.0})
    del A
    L, nup, ndown, t, J = 20, 9, 9, -1.0, 0.4
    lat = lambda v: square(5, 4, v, pbc=True)  # noqa: E731
    A = oracle.tj_csr(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4), force_complex=True)
    assert A.nrows == 9237800 and A.is_complex
    run("c4_tj_4x5_complex.json", "oracle (CPU) stored-CSR Lanczos of the t-J model on the periodic 5x4 cluster, 9 up 9 down, t=1, J=0.4, W=-J/4, complex<double>",
        A, True, {"L": L, "nup": nup, "ndown": ndown, "t": t, "J": J})


if __name__ == "__main__":
    main()
