#!/usr/bin/env python3
"""Regenerates heis_inf_temp.json from the reference's own stand-alone program
(src/HeisenbergInfiniteTemperatureEnergy.cpp, compiled by oracle/Makefile into oracle/_ref/heis_inf_temp
from the source where it lies under /root/reference).  Run in the build container only; the JSON
(inputs + expected outputs, no source text) is what travels and what the tests read.

Each record: args (L, twiceS, isPeriodic) -> the program's last output line "avg sum count":
count = number of states in the Sz=0-ish sector it enumerates (szPlusConst = twiceS*L/2),
sum = sum over those states of sum_bonds m_i m_j (nearest-neighbour chain, Jzz = 1)."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "..", "..", "oracle", "_ref", "heis_inf_temp")
CASES = [(4, 1, 0), (4, 1, 1), (8, 1, 0), (8, 1, 1), (10, 1, 1), (12, 1, 0), (12, 1, 1), (16, 1, 1),
         (4, 2, 1), (6, 2, 0), (6, 2, 1), (8, 2, 1), (4, 3, 1), (6, 3, 0)]
out = []
for L, twiceS, per in CASES:
    txt = subprocess.check_output([BIN, str(L), str(twiceS), str(per)], text=True)
    last = [l for l in txt.splitlines() if l and not l.startswith("#")][-1].split()
    out.append({"L": L, "twiceS": twiceS, "periodic": per, "avg": last[0], "sum": float(last[1]), "count": int(last[2])})
json.dump({"generator": "oracle/_ref/heis_inf_temp (reference src/HeisenbergInfiniteTemperatureEnergy.cpp)", "cases": out},
          open(os.path.join(HERE, "heis_inf_temp.json"), "w"), indent=1)
print("wrote", len(out), "cases")
