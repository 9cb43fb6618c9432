"""Shared test inputs: small lattices built independently of the product's geometry module."""
import numpy as np


def chain(L, v, pbc=False):
    m = np.zeros((L, L))
    for i in range(L - 1):
        m[i, i + 1] = m[i + 1, i] = v
    if pbc and L > 2:
        m[0, L - 1] = m[L - 1, 0] = v
    return m


def square(lx, ly, v, pbc=True):
    """lx x ly lattice, site = x*ly + y."""
    L = lx * ly
    m = np.zeros((L, L))
    for x in range(lx):
        for y in range(ly):
            s = x * ly + y
            for (dx, dy) in ((1, 0), (0, 1)):
                xx, yy = x + dx, y + dy
                if xx >= lx:
                    if not pbc or lx <= 2:
                        continue
                    xx = 0
                if yy >= ly:
                    if not pbc or ly <= 2:
                        continue
                    yy = 0
                t = xx * ly + yy
                m[s, t] = m[t, s] = v
    return m


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
