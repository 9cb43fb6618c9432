#!/usr/bin/env python3
"""bench.py -- Lanczos iterations/s and CSR-SpMV achieved HBM GB/s on MI355X.

A "step" is one Lanczos iteration (x += H y, a_j, x -= a_j y, b_j, swap/scale) on a Hamiltonian
that is already resident in HBM when the timed region starts.  N=1 workload: BASELINE.json
configs[1], 2-D Hubbard 4x4, 8 up 8 down, periodic, t=1, U=4 (N=165,636,900 rows,
Z=5,819,376,420 nnz, 71 GB CSR), assembled on the device.  N>1: the same matrix 1-D
row-partitioned over the ranks (strong scaling), all-gather of the Lanczos vector per step via
torch.distributed (backend nccl == RCCL over xGMI), one process per GPU.

  python bench.py --gpus 1 --steps 40 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured float4 copy is ~6290 GB/s
METRIC = "Lanczos iterations/sec + SpMV achieved HBM GB/s vs roofline, 1/2/4/8 GPU"
# environment switches that change the kernel or the layout: a committed PMC traffic figure only applies without them
LAYOUT_ENV = ("LPP_PB_CHAIN", "LPP_PB_SEG", "LPP_SPLIT_PANEL", "LPP_ONTHEFLY_KRON", "LPP_COMPRESS_VALUES", "LPP_SHARED_OFFSETS", "LPP_LOCAL16", "LPP_DIAG_CODES", "LPP_BLOCK_TEMPLATE", "LPP_SPMV_KERNEL",
              "LPP_WINDOW_ROWS", "LPP_K2_VARIANT", "LPP_KRON_NO_WINDOW", "LPP_KRON_NO_PACK", "LPP_TEMPLATE_PACK", "LPP_PRODUCT_LAYOUT")


ENV_SWITCHES_AT_START = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("LPP_") and k not in ("LPP_RCCL_ID_FILE", "LPP_RCCL_NONCE")}


def csrc_hash():
    """sha256 over the engine sources: profiles/traffic.json is stamped with it by scripts/traffic_stamp.py, so a
    PMC byte count is only ever quoted for the code it was measured on."""
    import hashlib
    d = os.path.join(ROOT, "lanczosplusplus_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(engine, name, world, spmv_kernel):
    """per-launch HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes, or None when the figure
    does not belong to this code / this configuration"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None, "no profiles/traffic.json"
    if tj.get("csrc_hash") != csrc_hash():
        return None, "stale: profiles/traffic.json was measured on other engine sources"
    if engine == "onthefly" and os.environ.get("LPP_ONTHEFLY_KRON") not in (None, "0") and not any(os.environ.get(k) is not None for k in LAYOUT_ENV if k != "LPP_ONTHEFLY_KRON"):
        engine = "onthefly_kron"  # the fused block-order kernel has a PMC pass of its own (profiles/*_c2otf_kron_*)
    elif world != 1 or spmv_kernel != 0 or any(os.environ.get(k) is not None for k in LAYOUT_ENV):
        return None, "not measured for this configuration"
    if world != 1 or spmv_kernel != 0:
        return None, "not measured for this configuration"
    ent = tj.get(engine, {}).get(name)
    if not ent:
        return None, "no PMC pass committed for this workload"
    return float(ent["traffic_bytes_per_launch"]), ent.get("source", "profiles/")


def square_lattice(lx, ly, v, pbc=True):
    L = lx * ly
    m = np.zeros((L, L))
    for x in range(lx):
        for y in range(ly):
            s = x * ly + y
            for dx, dy in ((1, 0), (0, 1)):
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


def chain(L, v, pbc=False):
    m = np.zeros((L, L))
    for i in range(L - 1):
        m[i, i + 1] = m[i + 1, i] = v
    if pbc and L > 2:
        m[0, L - 1] = m[L - 1, 0] = v
    return m


WORKLOADS = {
    # name: (model, params)
    "hubbard_4x4_half_filling_pbc_U4": ("hubbard", dict(L=16, nup=8, ndown=8, hop=lambda: square_lattice(4, 4, -1.0), U=4.0)),
    "hubbard_4x4_7up7down_pbc_U4": ("hubbard", dict(L=16, nup=7, ndown=7, hop=lambda: square_lattice(4, 4, -1.0), U=4.0)),
    # 2.36e9 states: 14x more than fits one GPU as a stored CSR (would be ~1 TB); matrix-free engine only
    "hubbard_3x6_half_filling_pbc_U4": ("hubbard", dict(L=18, nup=9, ndown=9, hop=lambda: square_lattice(3, 6, -1.0), U=4.0)),
    # 3.4e8 states, ~1.2e10 non-zeros (144 GB as a plain CSR): the largest 3x6 sector whose one-species space fits the LDS window
    # 1.34e8 states in 43758 blocks of 3060 positions: the chained step with a coupling panel (5.6 MB) beyond one XCD's L2
    "hubbard_3x6_4up8down_pbc_U4": ("hubbard", dict(L=18, nup=4, ndown=8, hop=lambda: square_lattice(3, 6, -1.0), U=4.0)),
    "hubbard_3x6_6up6down_pbc_U4": ("hubbard", dict(L=18, nup=6, ndown=6, hop=lambda: square_lattice(3, 6, -1.0), U=4.0)),
    # BASELINE config 5's lattice: 4x5, the (6,6) sector = 1.5e9 states (SURVEY 8(e) option 1); matrix-free engine only on one GPU
    "hubbard_4x5_6up6down_pbc_U4": ("hubbard", dict(L=20, nup=6, ndown=6, hop=lambda: square_lattice(4, 5, -1.0), U=4.0)),
    "hubbard_4x5_7up6down_pbc_U4": ("hubbard", dict(L=20, nup=7, ndown=6, hop=lambda: square_lattice(4, 5, -1.0), U=4.0)),  # 3.0e9 states
    # 6.0e9 states, 48 GB per vector, 77520 blocks of 77520 positions: more blocks than one LDS image of the coupling lists holds
    "hubbard_4x5_7up7down_pbc_U4": ("hubbard", dict(L=20, nup=7, ndown=7, hop=lambda: square_lattice(4, 5, -1.0), U=4.0)),
    "hubbard_4x5_8up7down_pbc_U4": ("hubbard", dict(L=20, nup=8, ndown=7, hop=lambda: square_lattice(4, 5, -1.0), U=4.0)),  # 9.8e9 states, 78 GB per vector
    "hubbard_chain_L12_half_filling_U4": ("hubbard", dict(L=12, nup=6, ndown=6, hop=lambda: chain(12, -1.0), U=4.0)),
    "hubbard_chain_L14_half_filling_U4": ("hubbard", dict(L=14, nup=7, ndown=7, hop=lambda: chain(14, -1.0), U=4.0)),
    # complex hoppings (Peierls phase on every bond): the reference's SolverOptions=useComplex path
    "hubbard_chain_L14_complex_U4": ("hubbard", dict(L=14, nup=7, ndown=7, hop=lambda: chain(14, -1.0) * np.where(np.triu(np.ones((14, 14)), 1) > 0, np.exp(0.2j), np.exp(-0.2j)), U=4.0)),
    # 6.4e7 states, complex: the largest 4x4 sector whose one-species space (8008 complex positions = 125 KB) fits the LDS window of the
    # product-basis layout for complex hoppings
    "hubbard_4x4_6up6down_complex_U4": ("hubbard", dict(L=16, nup=6, ndown=6, hop=lambda: square_lattice(4, 4, -1.0) * np.where(np.triu(np.ones((16, 16)), 1) > 0, np.exp(0.2j), np.exp(-0.2j)), U=4.0)),
    # 1.47e8 complex states (2.36 GB per vector): 12870 complex positions per row do not fit one LDS window -- the pieces form with four value groups
    "hubbard_4x4_8up7down_complex_U4": ("hubbard", dict(L=16, nup=8, ndown=7, hop=lambda: square_lattice(4, 4, -1.0) * np.where(np.triu(np.ones((16, 16)), 1) > 0, np.exp(0.2j), np.exp(-0.2j)), U=4.0)),
    "heisenberg_chain_L28_sz0_obc": ("heisenberg", dict(L=28, sz=14, j=1.0, pbc=False)),
    "heisenberg_chain_L24_sz0_obc": ("heisenberg", dict(L=24, sz=12, j=1.0, pbc=False)),
    "tj_4x5_9up9down_complex": ("tj", dict(L=20, nup=9, ndown=9, lx=5, ly=4, t=-1.0, j=0.4)),
    "tj_chain_L12_5up5down_complex": ("tj", dict(L=12, nup=5, ndown=5, lx=12, ly=1, t=-1.0, j=0.4)),
}


def assemble(engine, name, comm=None, onthefly=False):
    model, p = WORKLOADS[name]
    if model == "hubbard":
        L = p["L"]
        if onthefly:
            engine.setup_hubbard_onthefly(L, p["nup"], p["ndown"], p["hop"](), np.full(L, p["U"]), np.zeros(L), comm=comm)
            return
        engine.assemble_hubbard(L, p["nup"], p["ndown"], p["hop"](), np.full(L, p["U"]), np.zeros(L), comm=comm)
    elif model == "heisenberg":
        if comm is not None:
            raise SystemExit("multi-GPU bench is implemented for the Hubbard workloads")
        L = p["L"]
        engine.assemble_heisenberg(L, p["sz"], chain(L, p["j"], p["pbc"]), chain(L, p["j"], p["pbc"]))
    else:
        if comm is not None:
            raise SystemExit("multi-GPU bench is implemented for the Hubbard workloads")
        L = p["L"]
        lat = (lambda v: square_lattice(p["lx"], p["ly"], v, pbc=True)) if p["ly"] > 1 else (lambda v: chain(L, v))
        engine.assemble_tj(L, p["nup"], p["ndown"], lat(p["t"]), lat(p["j"]), lat(p["j"]), lat(-p["j"] / 4))


def cpu_baseline(name, nrows, budget_s=15.0):
    """Reference-style CPU path timed on this box's host cores on a bounded sample: the oracle's threaded
    on-the-fly Hubbard x += H y (HubbardHelper::matrixVectorProduct, the reference's only multi-core
    Hubbard path) over the first M rows, or the oracle's stored-CSR Lanczos iteration for the other models."""
    import oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a one-GPU box's CPU share is 16 cores; LPP_CPU_THREADS overrides
    cores = int(os.environ.get("LPP_CPU_THREADS", min(avail, 16)))
    model, p = WORKLOADS[name]
    if model == "hubbard":
        L = p["L"]
        hop, U, V = p["hop"](), np.full(L, p["U"]), np.zeros(L)
        y = oracle.fill_random(nrows, 99)
        x = np.zeros(nrows)
        rate = 0.0
        for m in (min(nrows, 100_000), min(nrows, 1_000_000)):  # warm-up, then calibration
            t0 = time.time()
            oracle.hubbard_otf_mvp(L, p["nup"], p["ndown"], hop, U, V, x, y, 0, m, cores)
            rate = m / max(time.time() - t0, 1e-9)
        m2 = int(min(nrows, max(1_000_000, rate * budget_s)))
        t0 = time.time()
        oracle.hubbard_otf_mvp(L, p["nup"], p["ndown"], hop, U, V, x, y, 0, m2, cores)
        dt = time.time() - t0
        # the three BLAS-1 passes of an iteration (dot; axpy + norm; swap/scale: SURVEY 8(d), 9 N s bytes) on the same m2 rows,
        # numpy on one core as the reference's LanczosSolver runs them (PsimagLite vectors, no threading)
        xs, ys = x[:m2], y[:m2].copy()
        t0 = time.time()
        a = float(ys @ xs)
        xs -= a * ys
        b = float(np.sqrt(xs @ xs)) or 1.0
        ys, xs = xs / b, -b * ys
        dt_blas = time.time() - t0
        its_spmv = (m2 / dt) / nrows
        its = (m2 / (dt + dt_blas)) / nrows
        # BASELINE.md section 3 single-thread leg (mirrors the reference's serial CrsMatrix loop), a tenth of the budget
        m1 = int(min(nrows, max(100_000, rate / max(cores, 1) * budget_s / 10)))
        t0 = time.time()
        oracle.hubbard_otf_mvp(L, p["nup"], p["ndown"], hop, U, V, x, y, 0, m1, 1)
        its1 = (m1 / max(time.time() - t0, 1e-9)) / nrows
        sample = ("whole iteration, extrapolated: on-the-fly x+=Hy (oracle port of HubbardHelper::matrixVectorProduct, HubbardHelper.h:105-134, %d threads) "
                  "plus the three BLAS-1 passes of the recurrence (numpy, one thread) timed over the first %d of %d rows (%.1f s + %.2f s) and scaled to all rows"
                  % (cores, m2, nrows, dt, dt_blas))
        return {"value": its, "unit": "iterations/s", "cores": cores, "kind": "port", "sample": sample, "rows_timed": m2,
                "what": "whole iteration (product + BLAS-1), extrapolated from a row sample", "spmv_only_value": its_spmv,
                "single_thread_value": its1, "single_thread_rows": m1, "single_thread_what": "x+=Hy only, one thread"}
    # other models: the reference's stored path (serial CrsMatrix::matrixVectorProduct) restated with OpenMP rows on the
    # oracle's own assembly of the same Hamiltonian, plus the three BLAS-1 passes of an iteration
    L = p["L"]
    if model == "heisenberg":
        A = oracle.heis_csr(L, 1, p["sz"], chain(L, p["j"], p["pbc"]), chain(L, p["j"], p["pbc"]))
    else:
        lat = (lambda v: square_lattice(p["lx"], p["ly"], v, pbc=True)) if p["ly"] > 1 else (lambda v: chain(L, v))
        A = oracle.tj_csr(L, p["nup"], p["ndown"], lat(p["t"]), lat(p["j"]), lat(p["j"]), lat(-p["j"] / 4), force_complex=True)
    y = oracle.fill_random(A.nrows, 99, A.is_complex)
    x = np.zeros_like(y)
    oracle.spmv_acc(A, x, y, cores)  # warm-up
    reps, t0 = 0, time.time()
    while time.time() - t0 < budget_s and reps < 50:
        oracle.spmv_acc(A, x, y, cores)
        a = np.vdot(y, x).real
        x -= a * y
        b = np.linalg.norm(x)
        x, y = -b * y, x / b
        reps += 1
    dt = time.time() - t0
    A1rows = A.nrows
    t1 = time.time()
    oracle.spmv_acc(A, x, y, 1)  # single-thread leg: one serial product
    dt1 = time.time() - t1
    return {"value": reps / dt, "unit": "iterations/s", "cores": cores, "kind": "port", "what": "full iterations on the whole matrix",
            "single_thread_spmv_s": dt1, "single_thread_rows": A1rows,
            "sample": "%d full Lanczos iterations (OpenMP stored-CSR x+=Hy from the oracle + numpy BLAS-1) on the same %d x %d matrix, %.1f s" % (reps, A.nrows, A.nrows, dt)}


GOLDEN = {"hubbard_4x4_half_filling_pbc_U4": "c2_hubbard4x4_U4.json",  # CPU-oracle Lanczos runs at full size (tests/golden/make_*.py)
          "heisenberg_chain_L28_sz0_obc": "c3_heisenberg_L28.json", "tj_4x5_9up9down_complex": "c4_tj_4x5_complex.json"}


def kernel_name(engine, layout):
    if engine == "onthefly" and (layout or {}).get("kernel") != "product":
        return "k_spmv_kron_packed / k_spmv_kron_chunked (matrix-free x += H y, fused a_j partial)"
    k = (layout or {}).get("kernel")
    if k == "product" and layout.get("segments", 0) > 0:
        # in-block matrix decomposed by the high sites of the basis word (lpp_pbseg.h); a one-block matrix (the spin chain) has no couplings
        return ("%sk_pb_up_seg + k_pb_combine (product-basis form H = 1(x)T + C(x)1 + D; T by %d high-site segments per row in %d LDS-window items: "
                "class-shared 16-bit low-low lists and cross-hop word tables, high-high hops as runs of the row in L2; the streaming pass forms x and applies "
                "the recurrence update: the launches are the WHOLE scale-free Lanczos step)"
                % ("" if layout.get("one_block") else "k_pb_down + ", layout["segments"], layout.get("pieces", 1)))
    if k == "product" and (layout.get("pieces", 1) > 1 or not layout.get("chained_step")):
        return ("k_pb_down%s + k_pb_up%s + k_pb_combine (product-basis form H = 1(x)T + C(x)1 + D; rows of %d LDS-window pieces, entries that leave a piece read "
                "from the row in L2; the streaming pass forms x and applies the recurrence update: the three launches are the WHOLE scale-free Lanczos step)"
                % ("_parts" if layout.get("coupling_parts", 1) > 1 else "", "_big" if layout.get("pieces", 1) > 1 else "", layout.get("pieces", 1)))
    return {"window": "k_spmv_window (stored matrix, LDS source window; x += H y, fused a_j partial)",
            "sliced": "k_spmv_sliced (stored matrix, wave-interleaved slices; x += H y, fused a_j partial)",
            "rowgroup": "k_spmv_rowgroup (plain CSR; x += H y, fused a_j partial)",
            "hole_major": ("k_tj_apply (t-J without a stored matrix: states ordered (hole configuration, spin pattern of the occupied sites); every row re-derives its "
                           "entries from its block's bonds and hole moves, patterns ranked through two LDS tables; x += H y, fused a_j partial)"),
            "product": ("k_pb_up<CHAIN> + k_pb_down<RMW> (product-basis form H = 1(x)T + C(x)1 + D: in-block part from the LDS window, block couplings "
                        "panel-wise from L2; the two launches are the WHOLE scale-free Lanczos step -- the previous step's axpy rides in k_pb_up)"
                        if (layout or {}).get("chained_step") else
                        "k_pb_up + k_pb_down (product-basis form H = 1(x)T + C(x)1 + D: in-block part from the LDS window, block couplings panel-wise from L2)")}.get(k, str(k))


def pb_chained():
    return os.environ.get("LPP_PB_CHAIN", "1") != "0"


def per_rank_bytes(eng, comm, st, esz, engine):
    """device memory one rank holds: matrix + 2 work vectors (+ exchange buffers)"""
    n = st["nrows"]
    b = 2.0 * n * esz
    if engine == "stored":
        lay = eng.layout(0)
        b += lay["resident_bytes"]
        if lay["kernel"] == 4:  # product-basis layout: the two parts of a product have buffers of their own (pb.u, pb.z)
            b += 2.0 * n * esz
        elif comm is not None:  # general layout on several ranks: the remote part is a matrix of its own
            b += eng.layout(1)["resident_bytes"]
    elif engine == "onthefly":
        try:
            lay = eng.layout(0)  # the product-basis form of the matrix-free engine
            b += lay["resident_bytes"] + 2.0 * n * esz
        except Exception:
            pass  # the fused block-order kernels: two one-species matrices, negligible
    if comm is not None:
        if hasattr(comm, "buffer_bytes"):  # the C-level communicator owns its buffers
            b += comm.buffer_bytes
        else:
            for t in (comm.send, comm.gath, comm.send2, comm.recv2):
                if t is not None:
                    b += t.numel() * 8
    return b


def generic_csr_leg(name, is_complex, device, iters=10):
    """x += H y with the plain 12-byte-per-entry layout (no value dictionary, no shared offsets): achieved = SURVEY 8(d)
    algorithmic bytes / HIP-event time of back-to-back launches"""
    from lanczosplusplus_amd import LanczosEngine
    saved = {k: os.environ.get(k) for k in ("LPP_COMPRESS_VALUES", "LPP_SHARED_OFFSETS", "LPP_PRODUCT_LAYOUT", "LPP_LOCAL16", "LPP_SPLIT_PANEL")}
    os.environ.update(LPP_COMPRESS_VALUES="0", LPP_SHARED_OFFSETS="0", LPP_PRODUCT_LAYOUT="0", LPP_LOCAL16="0")  # 8-byte values, 32-bit columns
    # (LPP_SPLIT_PANEL=1 would take the split-panel row order for the leaving entries: measured 8.1 + 9.9 ms in two launches, 19.3 ms
    #  back to back against 18.7 ms in one kernel -- DESIGN.md section 5 -- so the leg keeps the one-kernel form)
    try:
        with LanczosEngine(dtype="c128" if is_complex else "f64", device=device, max_steps=8, eps=0.0, save_vectors=0, compress_values=0) as e:
            assemble(e, name)
            lay, st = e.layout(0), e.stats()
            ms = e.bench_spmv(warmup=2, iters=iters)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    gbs = (st["spmv_bytes"] / 1e9) / (ms / 1e3)
    kern = {1: "k_spmv_rowgroup", 2: "k_spmv_sliced", 3: "k_spmv_window"}.get(lay["kernel"])
    if lay.get("split_panel"):
        kern += " (entries inside the row blocks, LDS window) + k_spmv_sliced (entries that leave them, rows panel-major: gathers from one XCD's L2)"
    return {"kernel": kern, "layout": "plain values and 32-bit columns (12 B per f64 entry)" + ("; split-panel row order for the leaving entries (+4 B per row)" if lay.get("split_panel") else ""),
            "spmv_ms": ms, "algorithmic_bytes_per_launch": st["spmv_bytes"], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "target_frac": 0.60, "resident_GB": round(lay["resident_bytes"] / 1e9, 2), "launches_timed": iters}


def reortho_leg(name, is_complex, device, onthefly, nrows, steps=24):
    """Blocked CGS2 against the on-device Krylov basis (north_star: "a blocked Gram-Schmidt"): `steps` Lanczos steps with
    LanczosOptions=reortho, every k_multi_dot / k_multi_axpy sweep of a step bracketed by HIP events on the engine's stream
    (lpp_stats.reortho_*).  Step j orthogonalises against j+1 columns, twice.  Bytes: the SURVEY 8(d) figure (per pass: the j+1
    columns read for the dots and again for the update, x read and written once = (2(j+1) + 2) N s) and what the blocked kernels
    move (panels of 8 columns: x is read once per panel by the dot sweep, read and written once per panel by the update)."""
    from lanczosplusplus_amd import LanczosEngine
    esz = 16 if is_complex else 8
    with LanczosEngine(dtype="c128" if is_complex else "f64", device=device, max_steps=steps + 2, eps=0.0, reortho=True, time_kernels=True) as e:
        assemble(e, name, None, onthefly=onthefly)
        e.begin(None)
        e.step(2)
        e.sync()
        s0 = e.stats()
        t0 = time.perf_counter()
        e.step(steps - 2)
        e.sync()
        dt = time.perf_counter() - t0
        s1 = e.stats()
        lay_rows = nrows
    calls = s1["reortho_calls"] - s0["reortho_calls"]
    ms = s1["reortho_ms_total"] - s0["reortho_ms_total"]
    cols = s1["reortho_columns"] - s0["reortho_columns"]
    panels = sum(-(-(j + 1) // 8) for j in range(2, steps))
    alg = 2.0 * (2.0 * cols + 2.0 * calls) * lay_rows * esz
    moved = 2.0 * (2.0 * cols + 3.0 * panels) * lay_rows * esz
    return {"what": "blocked CGS2 (k_multi_dot + k_multi_axpy, panels of 8 Krylov columns, two passes) on the engine's work vectors", "steps_timed": calls,
            "first_step": 2, "columns_total": cols, "ms_total": ms, "ms_per_step_avg": ms / max(calls, 1), "step_ms_with_reortho_avg": 1e3 * dt / max(steps - 2, 1),
            "algorithmic_bytes": alg, "achieved": (alg / 1e9) / (ms / 1e3) if ms > 0 else 0.0, "kernel_model_bytes": moved,
            "moved_GBps": (moved / 1e9) / (ms / 1e3) if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ((alg / 1e9) / (ms / 1e3) / HBM_PEAK_GBS) if ms > 0 else 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("LPP_BENCH_WORKLOAD", "hubbard_4x4_half_filling_pbc_U4"))
    ap.add_argument("--spmv-kernel", type=int, default=int(os.environ.get("LPP_BENCH_KERNEL", "0")))
    ap.add_argument("--engine", default=os.environ.get("LPP_BENCH_ENGINE", "stored"), choices=["stored", "onthefly"],
                    help="stored CSR (default, the BASELINE metric) or the matrix-free Hubbard product")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-generic-csr", action="store_true", help="skip the plain-CSR kernel leg (second roofline object)")
    ap.add_argument("--no-e0-check", action="store_true", help="skip the converged solve against the CPU-oracle fixture")
    ap.add_argument("--no-reortho-leg", action="store_true", help="skip the blocked Gram-Schmidt leg (24 steps with LanczosOptions=reortho)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: start the N ranks ourselves, as children of a parent that never touches the GPU (no torch
        # import, no HIP call before or after -- a process that has initialised the GPU must not exec, and this one does not
        # need to), relay rank 0's one JSON line and leave with the children's status.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, stdout=subprocess.PIPE)
        lines = [ln for ln in child.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
        raise SystemExit(child.returncode if child.returncode != 0 or lines else 1)

    # stdout carries ONE JSON line: everything libraries print there on their own (RCCL's version banner under the boxes'
    # NCCL_DEBUG=VERSION is a plain printf) goes to stderr; the line itself is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    from lanczosplusplus_amd import LanczosEngine, tridiag_lowest

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # under a launcher the launcher decides
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal on a one-GPU box: LPP_BENCH_BACKEND=gloo lets several ranks share cuda:0 (RCCL needs one GPU per rank)
    backend = os.environ.get("LPP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    name = args.workload
    model, p = WORKLOADS[name]
    is_complex = model == "tj" or "complex" in name
    max_steps = args.steps + args.warmup + 2

    comm = None
    attempts = [(None, None)]  # (exchange, communicator kind)
    if world > 1:
        import torch.distributed as dist
        from lanczosplusplus_amd.comm import RcclComm, TorchDistComm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        from math import comb as binom
        n_up, n_dn = binom(p["L"], p["nup"]), binom(p["L"], p["ndown"])
        stride = -(-n_dn // world) * n_up
        # exchange per step: all-gather of the vector (N/P per rank out, N in) or the transposition scheme (two all-to-alls of
        # N/P each: 2/P of the volume).  At 2 ranks both move the same bytes over the pair's one link, but the transposition
        # scheme runs the product-basis kernels (2.9 against 4.3 ms of compute per rank and step, tests/diagnostics/
        # rank_share_timing.py) and sets up in 0.03 s instead of 0.3 s: it is the first choice throughout.  LPP_EXCHANGE overrides.
        first = os.environ.get("LPP_EXCHANGE", "transpose")
        exchanges = [first] + (["allgather"] if first != "allgather" else [])
        # communicator: the C-level one over librccl (include/lpp_comm_rccl.h; collectives issued from C, nothing of Python between
        # the kernels of a step) when every rank has a GPU of its own, else / on any failure torch.distributed.  LPP_BENCH_COMM overrides.
        kinds = ["rccl_c", "torch"] if backend == "nccl" else ["torch"]
        if os.environ.get("LPP_BENCH_COMM"):
            kinds = [os.environ["LPP_BENCH_COMM"]]
        attempts = [(x, k) for k in kinds for x in exchanges]
    elif os.environ.get("LPP_BENCH_COMM") == "rccl_c":
        from lanczosplusplus_amd.comm import RcclComm
        attempts = [("allgather", "rccl_c")]  # one rank, one-rank communicator: exercises the plumbing on a one-GPU box

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def all_agree(ok):
        """every rank takes the same branch: the minimum of the ranks' flags"""
        if world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        return ok

    mem_used = [None]

    def close_all(st):
        """release whatever an attempt created: communicator first (include/lpp_comm_rccl.h), then the engine"""
        en, c = st.get("en"), st.get("c")
        if en is not None:
            try:
                en.sync()
            except Exception:
                pass
        if c is not None and hasattr(c, "close"):
            c.close()
        if en is not None:
            en.close()
        st.clear()

    def stages(exchange, kind, st):
        """An attempt in stages; the ranks compare notes (all_agree) after every stage, and only the stages marked COLLECTIVE
        enqueue anything another rank waits for -- a rank that fails alone (an allocation, a missing library) is noticed by
        all of them before anyone blocks inside a collective it will never join."""
        if world > 1:
            nd, nu, w = n_dn, n_up, world
        elif "ndown" in p and "nup" in p:
            nd, nu, w = __import__("math").comb(p["L"], p["ndown"]), __import__("math").comb(p["L"], p["nup"]), 1
        else:
            nd, nu, w = 0, 0, 1  # no species blocks (Heisenberg, t-J): only the one-rank form, no exchange buffer

        def create():  # local
            # up indices per rank rounded up to a multiple of 16 (= lpp_xchg_chunk): real Hubbard matrices then take the
            # product-basis kernels on both parts of the product
            st["chunk"] = ((-(-nd // w)) * ((-(-nu // w) + 15) // 16 * 16) if exchange == "transpose" else 0) if exchange is not None else 0
            if kind == "torch":
                st["c"] = TorchDistComm(stride, max_steps, is_complex, device=torch.device("cuda", local_rank), xchg_chunk=st["chunk"])
            c = st.get("c")
            strm = c.stream_handle if c is not None else None
            with (c.stream_context() if c is not None else __import__("contextlib").nullcontext()):
                st["en"] = LanczosEngine(dtype="c128" if is_complex else "f64", device=local_rank, max_steps=max_steps, eps=0.0,
                                         save_vectors=0, spmv_kernel=args.spmv_kernel, time_kernels=True, stream=strm)
            if kind == "rccl_c":
                RcclComm.lib()  # a missing library fails here, on this rank alone, before anything collective
                st["ident"] = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    st["ident"].copy_(torch.frombuffer(bytearray(RcclComm.unique_id()), dtype=torch.uint8))

        def join():  # COLLECTIVE (rccl_c only): the 128-byte RCCL id travels by torch.distributed, ncclCommInitRank joins the ranks
            if kind != "rccl_c":
                return
            if world > 1:
                import torch.distributed as dist
                dist.broadcast(st["ident"], 0)
            st_ = stride if world > 1 else (-(-nd // w)) * nu
            st["c"] = RcclComm(rank, world, bytes(st["ident"].cpu().numpy().tobytes()), local_rank, st["en"].stream_ptr(), st_, max_steps, is_complex, st["chunk"])

        def matrix():  # local: assembly issues no collective
            c, en = st.get("c"), st["en"]
            with (c.stream_context() if c is not None else __import__("contextlib").nullcontext()):
                free0 = torch.cuda.mem_get_info(local_rank)[0]
                t_a = time.time()
                assemble(en, name, c, onthefly=(args.engine == "onthefly"))
                en.sync()
                st["t_asm"] = time.time() - t_a
                # device memory the engine holds once the matrix is set (matrix + work vectors + product buffers), from the driver's
                # own accounting; the product-basis assembler allocates nothing it frees again, so this is also its peak
                mem_used[0] = (free0 - torch.cuda.mem_get_info(local_rank)[0]) / 1e9
                st["s0"] = en.stats()

        def warm():  # COLLECTIVE: self-test of every callback (checked on every rank), then the warm-up steps
            c, en = st.get("c"), st["en"]
            with (c.stream_context() if c is not None else __import__("contextlib").nullcontext()):
                if kind == "rccl_c":
                    c.selftest()
                en.begin(None)
                en.step(args.warmup)
                en.sync()

        return [create, join, matrix, warm]

    def coefficients_match(a, b):
        """the first Lanczos coefficients against the CPU-oracle fixture of the workload (same built-in start vector): a wrong
        exchange shows up here at once.  None when the workload has no fixture."""
        gold = os.path.join(ROOT, "tests", "golden", GOLDEN.get(name, ""))
        if name not in GOLDEN or not os.path.exists(gold):
            return None
        g = json.load(open(gold))
        n = min(len(a), len(g["a"]), 30)
        if n == 0:
            return None
        da = max(abs(a[k] - g["a"][k]) / max(abs(g["a"][k]), 1e-300) for k in range(n))
        db = max(abs(b[k] - g["b"][k]) / max(abs(g["b"][k]), 1e-300) for k in range(n))
        return {"steps_compared": n, "max_rel_diff": max(da, db), "source": "tests/golden/" + GOLDEN[name]}

    comm = eng = None
    coeff_check = None
    tried = []
    state = {}
    for exchange, kind in attempts:
        ok, err = True, None
        for stage in stages(exchange, kind, state):
            try:
                stage()
            except Exception as ex:  # e.g. a collective the backend lacks, an allocation that fails on one rank
                ok, err = False, ex
            ok = all_agree(ok)
            if not ok:
                break
        if ok:
            comm, eng, t_asm, st0 = state.get("c"), state["en"], state["t_asm"], state["s0"]
            try:
                ctx = comm.stream_context() if comm is not None else __import__("contextlib").nullcontext()
                with ctx:
                    eng.stats()  # drains the warmup SpMV event timings
                    w0 = eng.stats()
                    torch.cuda.synchronize()
                    barrier()
                    t0 = time.perf_counter()
                    eng.step(args.steps)
                    eng.sync()
                    torch.cuda.synchronize()
                    barrier()
                    t1 = time.perf_counter()
                    w1 = eng.stats()
                    a, b = eng.coeffs()
                # several ranks: this gate is mandatory -- a wrong exchange or a mis-ordered collective shows up in the first
                # coefficients at once, and the attempt is then abandoned on every rank
                coeff_check = coefficients_match(a, b) if rank == 0 else None
                if coeff_check is not None and not (coeff_check["max_rel_diff"] < 1e-6):
                    ok, err = False, RuntimeError("Lanczos coefficients differ from the CPU-oracle fixture: %r" % coeff_check)
            except Exception as ex:
                ok, err = False, ex
            ok = all_agree(ok)
        tried.append({"exchange": exchange, "comm": kind, "ok": ok})
        if ok:
            break
        if rank == 0:
            sys.stderr.write("bench: exchange %r over %r failed (%r); falling back\n" % (exchange, kind, err))
        close_all(state)
        comm = eng = None
    if eng is None:
        raise SystemExit("bench: could not set up the engine")
    comm_kind = kind

    elapsed = t1 - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([float(st0["nrows"]), float(st0["nnz"]), w1["spmv_bytes"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        nrows_g, nnz_g, bytes_g = int(tot[0].item()), int(tot[1].item()), float(tot[2].item())
    else:
        nrows_g, nnz_g, bytes_g = st0["nrows"], st0["nnz"], w1["spmv_bytes"]

    spmv_ms = (w1["spmv_ms_total"] - w0["spmv_ms_total"])
    launches = (w1["spmv_launches"] - w0["spmv_launches"])
    # per-rank SpMV time per step (multi-GPU: local + remote kernel of this rank)
    spmv_ms_per_step = spmv_ms / max(args.steps, 1)
    spmv_s = spmv_ms_per_step / 1e3
    esz = 16 if is_complex else 8
    csr_bytes = w1["spmv_bytes"]  # SURVEY 8(d) plain-CSR model of this rank's rows: Z(s+4) + (N+1)8 + 3Ns

    traffic, traffic_note = committed_traffic(args.engine, name, world, args.spmv_kernel)

    if rank == 0:
        e0 = float(tridiag_lowest(a, b[:-1] if len(b) > 1 else b, 1)[0]) if len(a) else float("nan")
        layout = None
        try:
            lay = eng.layout(0)  # the matrix-free engine describes a layout only where it is the product-basis one
        except Exception:
            lay = None
        if lay is not None and (args.engine == "stored" or lay["kernel"] == 4):
            layout = {"kernel": {1: "rowgroup", 2: "sliced", 3: "window", 4: "product", 5: "hole_major"}.get(lay["kernel"]), "value_codes": bool(lay["coded"]),
                      "pieces": lay["pieces"], "segments": lay.get("segments", 0), "one_block": bool(lay["kernel"] == 4 and lay["rows_per_block"] >= st0["nrows"]), "coupling_parts": lay["coupling_parts"], "coupling_rounds": lay.get("coupling_rounds", 1), "chained_step": bool(lay["chained_step"]), "rows_by_list_length": bool(lay.get("rows_by_list_length", 0)),
                      "local16_columns": bool(lay["local16"]), "block_template": lay["block_template"], "diagonal_codes": bool(lay["diagonal_codes"]), "per_row_entries": lay["per_row_entries"],
                      "shared_offset_entries": lay["shared_entries"], "resident_GB": round(lay["resident_bytes"] / 1e9, 2)}
            min_bytes = float(lay["stream_bytes"]) + 3.0 * st0["nrows"] * esz
            if lay["kernel"] == 4 and world == 1:
                # the timed launches are product AND recurrence update: a two-phase step (the reduction a_j sits between the
                # phases) cannot move less than  w = beta x + alpha H y (y, x in; w out)  +  x = w - g y (w, y in; x out)
                min_bytes = float(lay["stream_bytes"]) + 6.0 * st0["nrows"] * esz
            if world > 1 and args.engine == "stored":
                lay1 = eng.layout(1)
                min_bytes += float(lay1["stream_bytes"])
        else:
            min_bytes = 3.0 * st0["nrows"] * esz  # matrix-free: x in/out and y once; H_up / H_down are L2-resident
        # roofline of the dominant kernel (the SpMV): bytes that crossed the L2/fabric boundary per launch (rocprofv3 PMC
        # passes of this very code, profiles/) over the live HIP-event time.  Never above 1: when no fresh PMC figure
        # exists, the least traffic the resident layout allows (min_bytes) stands in, which can only under-state it.
        moved = traffic if traffic else min_bytes
        achieved = (moved / 1e9) / spmv_s if spmv_s > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "achieved_basis": "pmc_traffic" if traffic else "min_bytes", "traffic_note": traffic_note,
                    "min_bytes": min_bytes, "wasted": (traffic / min_bytes) if traffic else None,
                    "kernel": kernel_name(args.engine, layout), "spmv_ms": spmv_ms_per_step, "launches_timed": launches,
                    "covers": ("product + recurrence update (whole step)" if (layout or {}).get("kernel") == "product" and world == 1 else "product x += H y"),
                    # the plain-CSR figure of SURVEY 8(d) (12 B per entry for f64): what a kernel streaming the reference's
                    # CrsMatrix would have to sustain for this time -- a compression ratio times a bandwidth, NOT a roofline number
                    "csr_equivalent_bytes": csr_bytes, "csr_equivalent_GBps": (csr_bytes / 1e9) / spmv_s if spmv_s > 0 else 0.0}
        out = {
            "metric": METRIC,
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "c128" if is_complex else "f64",
            "data": "synthetic",
            "config": {"workload": name, "rows": nrows_g, "nnz": nnz_g, "parallelism": "1-D row partition x%d" % world,
                       "reortho": False, "assembly": "on-device", "assembly_s": round(t_asm, 3), "engine": args.engine,
                       # what the engine actually keeps: the matrix-free entry point builds the product-basis layout (T, C and ONE
                       # diagonal code per row: N bytes resident, counted in per_rank_memory_GB) wherever a species' row fits the LDS
                       # window; only the fused block-order kernels (LPP_ONTHEFLY_KRON=1, rows beyond the window, complex hoppings)
                       # keep nothing per row, as the reference's InternalProductOnTheFly does
                       "engine_effective": ("product-basis layout: T, C, one diagonal code per row" if (layout or {}).get("kernel") == "product" else
                                            ("matrix-free: two one-species matrices, nothing per row" if args.engine == "onthefly" else "stored CSR, " + str((layout or {}).get("kernel")) + " layout")),
                       "exchange": (("transpose" if comm.xchg_chunk > 0 else "allgather") if comm is not None else None),
                       "communicator": ({"rccl_c": "liblpp_comm_rccl.so (collectives issued from C)", "torch": "torch.distributed (%s)" % backend}.get(comm_kind)),
                       "attempts": tried, "coefficients_vs_cpu_oracle": coeff_check,
                       "per_rank_memory_GB": round(per_rank_bytes(eng, comm, st0, esz, args.engine) / 1e9, 2),
                       "device_memory_after_setup_GB": (round(mem_used[0], 2) if mem_used[0] is not None else None),
                       # the engine reads layout switches from the environment (DESIGN.md §6, table of switches): a line says which ones were
                       # set when it was made (none in the default run; LPP_VERBOSE / LPP_ENGINE_LIB are not layout switches but are listed too)
                       "env_switches": ENV_SWITCHES_AT_START,
                       "layout": layout},
            "roofline": roofline,
            "e0_after_steps": e0,
        }
    if world == 1 and rank == 0:
        # |E0(GPU) - E0(CPU)|: a converged solve against the CPU-oracle fixture of this workload, when one is committed
        gold = os.path.join(ROOT, "tests", "golden", GOLDEN.get(name, ""))
        if name in GOLDEN and os.path.exists(gold) and not args.no_e0_check:
            g = json.load(open(gold))
            if comm is not None and hasattr(comm, "close"):
                eng.sync()
                comm.close()
                comm = None
            eng.close()
            with LanczosEngine(dtype="c128" if is_complex else "f64", device=local_rank, max_steps=g["max_steps"], min_steps=g["min_steps"], eps=g["eps"],
                               save_vectors=0, seed=g["seed"], spmv_kernel=args.spmv_kernel) as e2:
                assemble(e2, name, None, onthefly=(args.engine == "onthefly"))
                t_s = time.time()
                a2, b2, _ = e2.decomposition()
                t_s = time.time() - t_s
            eg = float(tridiag_lowest(a2, b2[:-1], 1)[0])
            out["e0_check"] = {"e0_gpu": eg, "e0_cpu": g["e0"], "abs_diff": abs(eg - g["e0"]), "rel_diff": abs(eg - g["e0"]) / abs(g["e0"]),
                               "steps_gpu": len(a2), "steps_cpu": g["steps"], "solve_s": round(t_s, 3),
                               "cpu_source": "tests/golden/" + GOLDEN[name] + " (" + g["what"] + ", " + g["generator"] + ")"}
        # the generic (uncompressed, 12 B per entry) CSR kernel on the same matrix: the north_star's ">= 60 % of the HBM
        # roofline on the CSR SpMV" is about THIS kernel; its algorithmic bytes are the SURVEY 8(d) figure
        if args.engine == "stored" and not args.no_generic_csr:
            if comm is not None and hasattr(comm, "close"):
                if not eng.closed:
                    eng.sync()
                comm.close()
                comm = None
            eng.close()
            try:
                out["generic_csr"] = generic_csr_leg(name, is_complex, local_rank)
            except Exception as ex:
                out["generic_csr"] = {"error": repr(ex)}
        if not args.no_reortho_leg and nrows_g * esz * 28 < 200e9:  # 26 Krylov columns + work vectors must fit
            if comm is not None and hasattr(comm, "close"):
                if not eng.closed:
                    eng.sync()
                comm.close()
                comm = None
            eng.close()
            try:
                out["reortho"] = reortho_leg(name, is_complex, local_rank, args.engine == "onthefly", nrows_g)
            except Exception as ex:
                out["reortho"] = {"error": repr(ex)}
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(name, nrows_g, args.cpu_budget)
            except Exception as ex:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": repr(ex)}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    # the communicator goes before the engine whose stream it was ordered against (include/lpp_comm_rccl.h)
    if comm is not None and hasattr(comm, "close"):
        if not eng.closed:
            eng.sync()
        comm.close()
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
