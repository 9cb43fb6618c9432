"""Connection matrices and input files for the in-scope models.

The reference obtains couplings as geometry(i, orb_i, j, orb_j, term) from PsimagLite's Geometry
(absent here) configured by InputNg labels (TotalNumberOfSites=, NumberOfTerms=,
GeometryKind=chain|ladder, GeometryOptions=ConstantValues, `Connectors 1 v`, IsPeriodicX/Y=,
LadderLeg=).  These helpers rebuild the dense L x L matrix of one term for `chain` and `ladder`
[PsimagLite conventions restated from memory -- UNVERIFIED: ladder site numbering is
site = x*leg + y] and parse the `Label=value` / `Label n v1 .. vn` input format of
TestSuite/inputs/*.inp.  The engine itself only ever sees the matrices.
"""
import re

import numpy as np


def chain(L, value, periodic=False):
    m = np.zeros((L, L))
    for i in range(L - 1):
        m[i, i + 1] = m[i + 1, i] = value
    if periodic and L > 2:
        m[0, L - 1] = m[L - 1, 0] = value
    return m


def ladder(L, leg, value_x, value_y, periodic_x=False, periodic_y=False):
    """leg-leg ladder / 2-D lattice: site = x*leg + y, x along the legs."""
    if L % leg:
        raise ValueError("TotalNumberOfSites must be a multiple of LadderLeg")
    lx = L // leg
    m = np.zeros((L, L))

    def add(a, b, v):
        if a != b:
            m[a, b] += v
            m[b, a] += v

    for x in range(lx):
        for y in range(leg):
            s = x * leg + y
            if x + 1 < lx:
                add(s, (x + 1) * leg + y, value_x)
            elif periodic_x and lx > 2:
                add(s, y, value_x)
            if y + 1 < leg:
                add(s, x * leg + y + 1, value_y)
            elif periodic_y and leg > 2:
                add(s, x * leg, value_y)
    return m


def parse_input(text):
    """InputNg-style file -> dict.  Scalars `Label=value`; vectors/matrices `Label n v1 ... vn`
    (possibly spread over several lines); repeated labels (one `Connectors` per term) become lists."""
    out = {}
    toks = text.split()
    i = 0

    multi = set()

    def put(k, v):
        if k in out:
            if k not in multi:
                out[k] = [out[k]]
                multi.add(k)
            out[k].append(v)
        else:
            out[k] = v

    while i < len(toks):
        t = toks[i]
        if "=" in t:
            k, v = t.split("=", 1)
            put(k, v)
            i += 1
        elif re.match(r"^[A-Za-z_]", t) and i + 1 < len(toks) and re.match(r"^\d+$", toks[i + 1]):
            n = int(toks[i + 1])
            vals = [float(x) for x in toks[i + 2:i + 2 + n]]
            put(t, np.array(vals))
            i += 2 + n
        else:
            i += 1
    return out


def terms_from_input(inp):
    """Dense coupling matrix per Hamiltonian term from a parsed input (chain / ladder, ConstantValues)."""
    L = int(inp["TotalNumberOfSites"])
    nterms = int(inp["NumberOfTerms"])
    kind = inp.get("GeometryKind", "chain")
    conns = inp["Connectors"]
    if isinstance(conns, np.ndarray):
        conns = [conns]
    px = int(inp.get("IsPeriodicX", 0)) != 0
    py = int(inp.get("IsPeriodicY", 0)) != 0
    kinds = kind if isinstance(kind, list) else [kind] * nterms
    mats = []
    c = 0
    for t in range(nterms):
        if kinds[t] == "chain":
            mats.append(chain(L, conns[c][0], px))
            c += 1
        elif kinds[t] == "ladder":
            legs = inp.get("LadderLeg", 2)  # repeated labels (one per term) are consumed in order
            leg = int(legs[min(t, len(legs) - 1)] if isinstance(legs, list) else legs)
            mats.append(ladder(L, leg, conns[c][0], conns[c + 1][0], px, py))
            c += 2
        else:
            raise ValueError("unsupported GeometryKind " + str(kinds[t]))
    return mats
