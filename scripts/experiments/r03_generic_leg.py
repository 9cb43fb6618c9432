"""only the plain-format (12 B per entry) leg of bench.py on one workload: for rocprofv3 runs of the split-panel layout"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hubbard_4x4_half_filling_pbc_U4"
print(json.dumps(bench.generic_csr_leg(name, False, 0, iters=int(os.environ.get("ITERS", "6")))))
