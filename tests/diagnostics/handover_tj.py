import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from helpers import square
from lanczosplusplus_amd import LanczosEngine
L, nup, ndown, t, J = 20, 9, 9, -1.0, 0.4
lat = lambda v: square(5, 4, v, pbc=True)
with LanczosEngine(dtype="c128", max_steps=300, save_vectors=0) as e:
    t0 = time.time(); e.assemble_tj(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4)); print("assemble", time.time() - t0)
    t0 = time.time(); rp, ci, va = e.get_csr(); print("get_csr", time.time() - t0, len(va))
    t0 = time.time(); e.set_model_tj(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4)); e.set_csr(rp, ci, va); print("set_model+set_csr", time.time() - t0, e.layout()["kernel"])
    t0 = time.time(); e.set_csr(rp, ci, va); print("plain set_csr", time.time() - t0, e.layout()["kernel"])
