#!/usr/bin/env python3
"""Record the per-launch HBM traffic of a workload's SpMV kernel(s) in profiles/traffic.json, stamped with the hash of
the engine sources it was measured on (bench.py quotes a figure only while that hash still matches).

  python scripts/traffic_stamp.py <engine: stored|onthefly> <workload> <pmc_summary.csv> <kernel substring>[,<second kernel>...]

pmc_summary.csv is what scripts/profile_round.sh writes: Kernel,Counter,Dispatches,MeanPerDispatch from separate rocprofv3
--pmc passes.  Bytes per launch = TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32 (exact request counts; FETCH_SIZE is half of a
wide read stream on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE KB*1024, summed over the named kernels (a product that is split
into two kernels moves the sum per step).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_hash  # noqa: E402


def main():
    engine, workload, path, kernels = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4].split(",")
    rows = list(csv.DictReader(open(path)))
    total_r = total_w = 0.0
    names = []
    for k in kernels:
        sel = [r for r in rows if k in r["Kernel"]]
        if not sel:
            raise SystemExit("no kernel matching %r in %s" % (k, path))
        kn = sorted({r["Kernel"] for r in sel})
        if len(kn) > 1:
            raise SystemExit("kernel substring %r is ambiguous: %s" % (k, kn))
        c = {r["Counter"]: float(r["MeanPerDispatch"]) for r in sel}
        total_r += c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
        total_w += c.get("WRITE_SIZE", 0) * 1024
        names.append(kn[0])
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    h = csrc_hash()
    try:
        tj = json.load(open(tj_path))
    except Exception:
        tj = {}
    if tj.get("csrc_hash") != h:
        tj = {"csrc_hash": h}  # figures measured on other sources are dropped, not carried along
    tj.setdefault(engine, {})[workload] = {"kernel": " + ".join(names), "traffic_bytes_per_launch": total_r + total_w,
                                            "read_bytes_exact_ea_requests": total_r, "write_bytes": total_w,
                                            "source": "profiles/" + os.path.basename(path) + " (rocprofv3 --pmc, separate passes)"}
    json.dump(tj, open(tj_path, "w"), indent=1)
    print("traffic.json[%s][%s] = %.3f GB per launch (hash %s)" % (engine, workload, (total_r + total_w) / 1e9, h))


if __name__ == "__main__":
    main()
