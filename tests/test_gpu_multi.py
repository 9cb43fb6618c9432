"""-m gpu: the row-partitioned multi-rank Lanczos loop on real HIP kernels.

A one-GPU box cannot host two RCCL ranks, so two processes share cuda:0 and exchange through the
gloo backend (same communicator class and callbacks as the nccl path; only the backend differs).
Checks: device-assembled local/remote CSRs are bit-identical to the host split of the oracle CSR,
energies match the oracle to 1e-10, coefficients match a single-rank run, host-CSR partition upload
(complex t-J) works, reortho + Ritz vectors work on slices.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(target, world, timeout=300):
    """Spawn `world` ranks; never leave a hung child behind (a deadlocked collective must fail the test, not the run).
    Every all-reduce of every communicator is recorded on the engine's stream (LPP_COMM_RECORD, lanczosplusplus_amd/comm.py)
    and checked on all ranks afterwards: a rank whose kernels saw anything but the sum of the partials fails the test with
    the call, the rank and the values on record."""
    os.environ["LPP_COMM_RECORD"] = "1"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in procs:
            r, out = q.get(timeout=timeout)
            res[r] = out
    finally:
        for p in procs:
            p.join(timeout=20)
        for p in procs:
            if p.is_alive():
                p.kill()
    for r, o in sorted(res.items()):
        assert not o.get("findings"), "rank %d: recorded all-reduces are wrong: %r" % (r, o["findings"][:4])
    return res


def _audit(comm, out):
    """collective: every rank checks the all-reduces this communicator recorded"""
    out.setdefault("findings", []).extend(comm.verify_record())
    out["recorded"] = out.get("recorded", 0) + comm.calls["allreduce"]


def _worker(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import chain
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        out = {}

        # ---- Hubbard, device assembly of the rank's row block ----
        L, nup, ndown = 10, 5, 4
        hop, U = chain(L, -1.0, True), np.full(L, 4.0)
        A = oracle.hubbard_csr(L, nup, ndown, hop, U)
        n_up = 252
        starts = lp.partition_rows(A.nrows, world, n_up)
        stride = int(-(-210 // world) * n_up)
        comm = TorchDistComm(stride, 200, False, device=dev)
        with comm.stream_context():
            e = lp.LanczosEngine(max_steps=200, stream=comm.stream_handle)
            e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm)
            lo, hi = int(starts[rank]), int(starts[rank + 1])
            rp = A.rowptr[lo:hi + 1] - A.rowptr[lo]
            (rpl, cl, vl), (rpr, cr, vr) = lp.split_csr(rank, world, starts, stride, rp, A.colind[A.rowptr[lo]:A.rowptr[hi]],
                                                        A.values[A.rowptr[lo]:A.rowptr[hi]])
            g = e.get_csr(0)
            out["loc_exact"] = bool(np.array_equal(g[0], rpl) and np.array_equal(g[1], cl) and np.array_equal(g[2], vl))
            g = e.get_csr(1)
            out["rem_exact"] = bool(np.array_equal(g[0], rpr) and np.array_equal(g[1], cr) and np.array_equal(g[2], vr))
            eg, zg, st = e.lanczos(1, want_vectors=True)
            out["e_hub"], out["steps_hub"] = float(eg[0]), st["steps"]
            # the slices of the Ritz vector, gathered: residual of the full vector
            zs = [None] * world
            dist.all_gather_object(zs, zg[0])
            z = np.concatenate(zs)
            r = oracle.spmv_acc(A, np.zeros_like(z), z) - eg[0] * z
            out["resid"] = float(np.linalg.norm(r))
            a, b, _ = e.decomposition()
            out["a"], out["b"] = a, b
            # matrix-free engine on the same partition
            e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm)
            ek, _, stk = e.lanczos(1, want_vectors=False)
            out["e_kron"], out["steps_kron"] = float(ek[0]), stk["steps"]
            e.close()
        _audit(comm, out)
        eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False)
        out["e_hub_oracle"], out["steps_oracle"] = float(eo[0]), so

        # ---- complex t-J, host CSR handed over as a row partition, with reortho ----
        L = 8
        T = oracle.tj_csr(L, 3, 3, chain(L, -1.0), chain(L, 0.4), chain(L, 0.4), chain(L, -0.1), force_complex=True)
        starts = lp.partition_rows(T.nrows, world, 1)
        stride = int(max(np.diff(starts)))
        comm2 = TorchDistComm(stride, 120, True, device=dev)
        with comm2.stream_context():
            e = lp.LanczosEngine(dtype="c128", max_steps=120, reortho=True, eps=1e-13, stream=comm2.stream_handle)
            lo, hi = int(starts[rank]), int(starts[rank + 1])
            e.set_csr_partition(comm2, T.nrows, starts, T.rowptr[lo:hi + 1] - T.rowptr[lo], T.colind[T.rowptr[lo]:T.rowptr[hi]],
                                T.values[T.rowptr[lo]:T.rowptr[hi]])
            init = oracle.fill_random(T.nrows, 77, True)
            eg, _, st = e.lanczos(2, init=init[lo:hi].copy(), want_vectors=False)
            out["e_tj"] = [float(v) for v in eg]
            e.close()
        _audit(comm2, out)
        out["e_tj_dense"] = [float(v) for v in np.linalg.eigvalsh(T.to_scipy().toarray())[:2]]
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_two_ranks_share_one_gpu_over_gloo():
    res = _run_ranks(_worker, 2)
    for r in (0, 1):
        assert "error" not in res[r], res[r].get("error")
    for r in (0, 1):
        o = res[r]
        assert o["loc_exact"] and o["rem_exact"]
        assert abs(o["e_hub"] - o["e_hub_oracle"]) <= 1e-10 * abs(o["e_hub_oracle"])
        assert o["steps_hub"] == o["steps_oracle"]
        assert o["resid"] < 1e-5
        assert abs(o["e_kron"] - o["e_hub_oracle"]) <= 1e-10 * abs(o["e_hub_oracle"]) and o["steps_kron"] == o["steps_oracle"]
        assert abs(o["e_tj"][0] - o["e_tj_dense"][0]) <= 1e-10 * abs(o["e_tj_dense"][0])
        assert abs(o["e_tj"][1] - o["e_tj_dense"][1]) <= 1e-8
    assert np.array_equal(res[0]["a"], res[1]["a"]) and np.array_equal(res[0]["b"], res[1]["b"])
    # single-rank run of the same matrix gives the same coefficients up to summation order
    import oracle
    from helpers import chain
    from lanczosplusplus_amd import LanczosEngine
    with LanczosEngine() as e:
        e.assemble_hubbard(10, 5, 4, chain(10, -1.0, True), np.full(10, 4.0))
        a1, b1, _ = e.decomposition()
    n = min(len(a1), len(res[0]["a"]))
    assert np.abs(a1[:n] - res[0]["a"][:n]).max() < 1e-8


def _worker_uneven(rank, world, port, q):
    """4 ranks, shards of unequal size (210 down-configurations -> 53,53,53,51), stored and matrix-free engines."""
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import chain
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        L, nup, ndown = 10, 5, 4
        hop, U = chain(L, -1.0, True), np.linspace(3.0, 5.0, L)
        n_up = 252
        stride = int(-(-210 // world) * n_up)
        comm = TorchDistComm(stride, 200, False, device=dev)
        out = {}
        with comm.stream_context():
            e = lp.LanczosEngine(max_steps=200, stream=comm.stream_handle)
            e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm)
            out["rows"] = e.rows()
            eg, _, st = e.lanczos(1, want_vectors=False)
            out["e_stored"], out["steps_stored"] = float(eg[0]), st["steps"]
            e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm)
            eg, zg, st = e.lanczos(2, want_vectors=True)  # keeps vectors: normalised recurrence on the matrix-free product
            out["e_kron"], out["steps_kron"] = [float(v) for v in eg], st["steps"]
            e.close()
        _audit(comm, out)
        # transposition exchange: same matrix, two all-to-alls per step instead of the all-gather
        per, peru = -(-210 // world), -(-n_up // world)
        comm_t = TorchDistComm(stride, 200, False, device=dev, xchg_chunk=per * peru)
        with comm_t.stream_context():
            e = lp.LanczosEngine(max_steps=200, stream=comm_t.stream_handle)
            e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm_t)
            ar0 = comm_t.calls["allreduce"]
            eg, _, st = e.lanczos(1, want_vectors=False)  # scale-free recurrence
            out["e_tx"], out["steps_tx"] = float(eg[0]), st["steps"]
            out["ar_tx"] = comm_t.calls["allreduce"] - ar0
            eg, zg, st = e.lanczos(1, want_vectors=True)  # normalised recurrence, Krylov basis kept
            out["e_tx2"] = float(eg[0])
            zs = [None] * world
            dist.all_gather_object(zs, zg[0])
            out["z_tx"] = np.concatenate(zs)
            out["xchg_calls"] = comm_t.calls["exchange"]
            # matrix-free engine with the same exchange: no buffer of the full vector length anywhere
            e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm_t)
            eg, _, st = e.lanczos(1, want_vectors=False)
            out["e_kron_tx"], out["steps_kron_tx"] = float(eg[0]), st["steps"]
            e.close()
        _audit(comm_t, out)
        A = oracle.hubbard_csr(L, nup, ndown, hop, U)
        eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False)
        out["e_oracle"], out["steps_oracle"] = float(eo[0]), so
        # complex hoppings, both index ranges uneven (126 = 32+32+32+30), transposition exchange
        L2 = 9
        hc = chain(L2, -1.0, True).astype(complex)
        hc[0, 1] *= np.exp(0.4j)
        hc[1, 0] = np.conj(hc[0, 1])
        U2 = np.full(L2, 3.0)
        per = peru = -(-126 // world)
        comm_c = TorchDistComm(per * 126, 200, True, device=dev, xchg_chunk=per * peru)
        with comm_c.stream_context():
            e = lp.LanczosEngine(dtype="c128", max_steps=200, stream=comm_c.stream_handle)
            e.assemble_hubbard(L2, 4, 4, hc, U2, comm=comm_c)
            out["rows_c"] = e.rows()
            eg, _, st = e.lanczos(1, want_vectors=False)
            out["e_tx_c"], out["steps_tx_c"] = float(eg[0]), st["steps"]
            e.setup_hubbard_onthefly(L2, 4, 4, hc, U2, comm=comm_c)
            eg, _, st = e.lanczos(1, want_vectors=False)
            out["e_kron_tx_c"] = float(eg[0])
            e.close()
        _audit(comm_c, out)
        Ac = oracle.hubbard_csr(L2, 4, 4, hc, U2)
        ec, _, sc = oracle.lanczos_solve(Ac, oracle.fill_random(Ac.nrows, 1234, True), want_vectors=False)
        out["e_oracle_c"], out["steps_oracle_c"] = float(ec[0]), sc
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_four_ranks_uneven_shards():
    world = 4
    res = _run_ranks(_worker_uneven, world)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
    assert [res[r]["rows"] for r in range(world)] == [53 * 252, 53 * 252, 53 * 252, 51 * 252]
    for r in range(world):
        o = res[r]
        assert abs(o["e_stored"] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"]) and o["steps_stored"] == o["steps_oracle"]
        assert abs(o["e_kron"][0] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"])
        assert abs(o["e_tx"] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"]) and o["steps_tx"] == o["steps_oracle"]
        assert abs(o["e_tx2"] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"])
        assert o["xchg_calls"] > 0
        # a_j and b_j^2 travel in ONE all-reduce per step; the few extra calls are the start vector's norm
        # and the steps already in flight when the lagged convergence check fires -- two per step would be >= 148 here
        assert o["steps_tx"] <= o["ar_tx"] <= o["steps_tx"] + 8, (o["ar_tx"], o["steps_tx"])
        assert abs(o["e_kron_tx"] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"]) and o["steps_kron_tx"] == o["steps_oracle"]
        assert abs(o["e_tx_c"] - o["e_oracle_c"]) <= 1e-10 * abs(o["e_oracle_c"]) and o["steps_tx_c"] == o["steps_oracle_c"]
        assert abs(o["e_kron_tx_c"] - o["e_oracle_c"]) <= 1e-10 * abs(o["e_oracle_c"])
    assert [res[r]["rows_c"] for r in range(world)] == [32 * 126, 32 * 126, 32 * 126, 30 * 126]
    assert all(res[r]["e_kron"] == res[0]["e_kron"] for r in range(world))
    # the Ritz vector assembled from the four slices is an eigenvector of the full matrix
    import oracle
    from helpers import chain
    A = oracle.hubbard_csr(10, 5, 4, chain(10, -1.0, True), np.linspace(3.0, 5.0, 10))
    z = res[0]["z_tx"]
    r = oracle.spmv_acc(A, np.zeros_like(z), z) - res[0]["e_tx2"] * z
    assert np.linalg.norm(r) < 1e-5


def _worker_beyond_lds(rank, world, port, q):
    """2 ranks, matrix-free engine with N_up = C(18,9) = 48620 beyond the LDS window (source row staged in three pieces),
    all-gather and transposition exchange."""
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import square
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        L, nup, ndown = 18, 9, 1
        hop, U = square(3, 6, -1.0, True), np.full(L, 4.0)
        n_up, n_dn = 48620, 18
        per, peru = -(-n_dn // world), -(-n_up // world)
        out = {}
        for name, chunk in (("allgather", 0), ("transpose", per * peru)):
            comm = TorchDistComm(per * n_up, 300, False, device=dev, xchg_chunk=chunk)
            with comm.stream_context():
                e = lp.LanczosEngine(max_steps=300, stream=comm.stream_handle)
                e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm)
                eg, _, st = e.lanczos(1, want_vectors=False)
                out["e_" + name], out["steps_" + name] = float(eg[0]), st["steps"]
                e.close()
            _audit(comm, out)
        if rank == 0:
            A = oracle.hubbard_csr(L, nup, ndown, hop, U)
            eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False, max_steps=300)
            out["e_oracle"], out["steps_oracle"] = float(eo[0]), so
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_two_ranks_matrix_free_beyond_lds(monkeypatch):
    monkeypatch.setenv("LPP_PB_SEG", "1")  # the in-block matrix decomposed by the high sites (taken by itself from 65536 positions on) on the exchange too
    world = 2
    res = _run_ranks(_worker_beyond_lds, world, timeout=400)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
    eo, so = res[0]["e_oracle"], res[0]["steps_oracle"]
    for r in range(world):
        for name in ("allgather", "transpose"):
            assert abs(res[r]["e_" + name] - eo) <= 1e-10 * abs(eo), (name, res[r], eo)
            assert abs(res[r]["steps_" + name] - so) <= 1


def _worker_4x5(rank, world, port, q):
    """BASELINE config 5's lattice (4x5, periodic, U = 4) over 4 ranks: the (3,3) sector (1140^2 = 1,299,600 states) through the
    transposition exchange, stored and matrix-free engines, against the oracle; plus the (4,3) sector (unequal species:
    4845 x 1140) matrix-free against its exact free-fermion energy."""
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import square
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        L = 20
        hop, U = square(4, 5, -1.0, True), np.full(L, 4.0)
        out = {}
        n_up = n_dn = 1140
        per, peru = -(-n_dn // world), -(-n_up // world)
        comm = TorchDistComm(per * n_up, 300, False, device=dev, xchg_chunk=per * peru)
        with comm.stream_context():
            e = lp.LanczosEngine(max_steps=300, stream=comm.stream_handle)
            e.assemble_hubbard(L, 3, 3, hop, U, comm=comm)
            out["rows"] = e.rows()
            a, b, _ = e.decomposition()
            out["a_st"], out["b_st"] = a, b
            e.setup_hubbard_onthefly(L, 3, 3, hop, U, comm=comm)
            a, b, _ = e.decomposition()
            out["a_mf"], out["b_mf"] = a, b
            out["xchg_calls"] = comm.calls["exchange"]
            e.close()
        _audit(comm, out)
        if rank == 0:
            A = oracle.hubbard_csr(L, 3, 3, hop, U)
            so, ao, bo, _, hist = oracle.lanczos_decomposition(A, oracle.fill_random(A.nrows, 1234), max_steps=300, nthreads=0)
            out["a_o"], out["b_o"], out["e_o"] = ao, bo, float(hist[-1])
        # unequal species, free fermions
        n_up, n_dn = 4845, 1140
        per, peru = -(-n_dn // world), -(-n_up // world)
        comm2 = TorchDistComm(per * n_up, 300, False, device=dev, xchg_chunk=per * peru)
        with comm2.stream_context():
            e = lp.LanczosEngine(max_steps=300, eps=1e-11, stream=comm2.stream_handle)
            e.setup_hubbard_onthefly(L, 4, 3, hop, np.zeros(L), comm=comm2)
            eg, _, st = e.lanczos(1, want_vectors=False)
            out["e_ff"] = float(eg[0])
            e.close()
        _audit(comm2, out)
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_four_ranks_config5_lattice_transposition_exchange():
    world = 4
    res = _run_ranks(_worker_4x5, world, timeout=400)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
    from helpers import rel, square
    from lanczosplusplus_amd import tridiag_lowest
    assert [res[r]["rows"] for r in range(world)] == [285 * 1140] * 4
    ao, bo, eo = res[0]["a_o"], res[0]["b_o"], res[0]["e_o"]
    lev = np.sort(np.linalg.eigvalsh(square(4, 5, -1.0, True)))
    exact = lev[:4].sum() + lev[:3].sum()
    for r in range(world):
        o = res[r]
        assert o["xchg_calls"] > 0
        for tag in ("st", "mf"):
            a, b = o["a_" + tag], o["b_" + tag]
            e0 = tridiag_lowest(a, b[:-1], 1)[0]
            assert abs(e0 - eo) <= 1e-10 * abs(eo), (tag, e0, eo)
            assert abs(len(a) - len(ao)) <= 1
            assert rel(a[:40], ao[:40]) < 1e-8 and rel(b[:40], bo[:40]) < 1e-8
            assert np.array_equal(a, res[0]["a_" + tag])  # every rank takes bitwise-identical decisions
        assert abs(o["e_ff"] - exact) <= 1e-10 * abs(exact)


@pytest.mark.parametrize("worker", ["uneven", "pb_tx"])
def test_four_ranks_stream_ordered_collectives_are_recorded_and_correct(worker, monkeypatch):
    """The communicators a multi-GPU node runs (torch.distributed over nccl, liblpp_comm_rccl.so) are stream-ordered: nothing
    waits on the host between a collective and the kernels that consume it.  A one-GPU box cannot run them with more than one
    rank, so the gloo emulation is switched to its stream-ordered form here (device tensors handed to gloo, which forks and
    joins the engine's stream with events exactly as ProcessGroupNCCL does) and every all-reduce is recorded on the engine's
    stream: the partial that went in and the value the next kernel reads.  _run_ranks fails on any record that is not the
    sum of the partials or differs between ranks; energies and coefficients are checked as in the host-staged runs."""
    monkeypatch.setenv("LPP_GLOO_STREAM_ORDERED", "1")
    world = 4
    res = _run_ranks(_worker_uneven if worker == "uneven" else _worker_pb_tx, world, timeout=400)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
        assert res[r]["recorded"] > 100  # the record is not empty
        o = res[r]
        if worker == "uneven":
            for key in ("e_stored", "e_tx", "e_tx2", "e_kron_tx"):
                assert abs(o[key] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"]), (key, o[key], o["e_oracle"])
            assert abs(o["e_kron"][0] - o["e_oracle"]) <= 1e-10 * abs(o["e_oracle"])
            assert abs(o["e_tx_c"] - o["e_oracle_c"]) <= 1e-10 * abs(o["e_oracle_c"])
        else:
            for tag in ("disorder", "ladder", "chain"):
                eo = res[0][tag + "_eo"]
                assert abs(o[tag + "_e"] - eo) <= 1e-10 * abs(eo) and abs(o[tag + "_e2"] - eo) <= 1e-10 * abs(eo) and abs(o[tag + "_mf_e"] - eo) <= 1e-10 * abs(eo)
                assert np.array_equal(o[tag + "_a"], res[0][tag + "_a"]) and np.array_equal(o[tag + "_b"], res[0][tag + "_b"])


def test_c_level_rccl_communicator_at_world_size_one():
    """include/lpp_comm_rccl.h: the lpp_comm a non-Python host uses (ncclAllGather / ncclAllReduce / grouped send-recv on HIP
    streams).  One rank only here (a one-GPU box cannot host two RCCL ranks): every callback runs once through
    lpp_rccl_comm_selftest for the all-gather and the transposition layout, f64 and complex, and an engine created on the
    communicator's stream with that lpp_comm solves a problem."""
    import ctypes as C
    import oracle
    from helpers import chain
    from lanczosplusplus_amd import LanczosEngine, _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "lanczosplusplus_amd", "csrc", "liblpp_comm_rccl.so"))
    lib.lpp_rccl_last_error.restype = C.c_char_p
    lib.lpp_rccl_comm_get.restype = C.POINTER(_capi.Comm)
    lib.lpp_rccl_comm_get.argtypes = [C.c_void_p]
    lib.lpp_rccl_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                         C.c_int32, C.c_int64]
    lib.lpp_rccl_comm_destroy.argtypes = [C.c_void_p]
    lib.lpp_rccl_comm_selftest.argtypes = [C.c_void_p]

    def ok(rc):
        assert rc == 0, lib.lpp_rccl_last_error().decode()

    import torch
    stream = torch.cuda.Stream(device=0)
    L, nup, ndown = 8, 4, 3
    n_up, n_dn = 70, 56
    for is_complex in (0, 1):
        for chunk in (0, n_dn * n_up):
            ident = C.create_string_buffer(128)
            ok(lib.lpp_rccl_unique_id(ident))
            h = C.c_void_p()
            ok(lib.lpp_rccl_comm_create(C.byref(h), 0, 1, ident, 0, C.c_void_p(stream.cuda_stream), n_dn * n_up, 200, is_complex, chunk))
            ok(lib.lpp_rccl_comm_selftest(h))
            cs = lib.lpp_rccl_comm_get(h)
            assert cs.contents.nranks == 1 and cs.contents.shard_stride == n_dn * n_up and cs.contents.red_len >= 6 * 202 + 8
            assert cs.contents.send_buf % 16 == 0 and cs.contents.gath_buf % 16 == 0
            if not is_complex:
                hop, U = chain(L, -1.0, True), np.full(L, 4.0)
                A = oracle.hubbard_csr(L, nup, ndown, hop, U)
                eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False)
                with torch.cuda.stream(stream):
                    e = LanczosEngine(max_steps=200, stream=C.c_void_p(stream.cuda_stream))

                    class Holder:  # what engine.py expects of a communicator object
                        struct = cs.contents
                    e.assemble_hubbard(L, nup, ndown, hop, U, comm=Holder())
                    eg, _, st = e.lanczos(1, want_vectors=False)
                    e.close()
                assert abs(eg[0] - eo[0]) <= 1e-10 * abs(eo[0]) and st["steps"] == so
            ok(lib.lpp_rccl_comm_destroy(h))


def _worker_pb_tx(rank, world, port, q):
    """Product-basis kernels on the transposition exchange: the in-block kernel on the rank's own down configurations, the
    panel-major coupling kernel on the received transposed slice (up range per rank rounded up to 16 = lpp_xchg_chunk), uneven
    shards (495 = 124+124+124+123 down configurations, 924 up indices in ranges of 240), scale-free and vector-keeping runs."""
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        os.environ["LPP_PRODUCT_LAYOUT"] = "1"  # below the size from which the layout is chosen by itself
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import chain, square
        from lanczosplusplus_amd._capi import lib
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        out = {}
        for tag, (L, nup, ndown, hop, U, ninj) in (("disorder", (12, 6, 5, chain(12, -1.0, True), np.random.default_rng(5).uniform(1, 5, 12), None)),  # plain-stream diagonal
                                                   ("ladder", (12, 6, 4, square(2, 6, -1.0, True), np.full(12, 4.0), None)),
                                                   ("chain", (12, 6, 6, chain(12, -1.0, False), np.where(np.arange(12) % 2 == 0, 3.0, 5.0),
                                                              chain(12, 0.5, False)))):  # HubbardOneBandExtended: Coulomb term in the diagonal codes
            from math import comb
            n_up, n_dn = comb(L, nup), comb(L, ndown)
            per = -(-n_dn // world)
            chunk = lib().lpp_xchg_chunk(n_up, n_dn, world)
            assert chunk % per == 0 and (chunk // per) % 16 == 0 and (chunk // per) * world >= n_up
            comm = TorchDistComm(per * n_up, 300, False, device=dev, xchg_chunk=chunk)
            with comm.stream_context():
                e = lp.LanczosEngine(max_steps=300, stream=comm.stream_handle)
                e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm, ninj=ninj)
                out[tag + "_kernel"] = e.layout(0)["kernel"]
                out[tag + "_rounds"] = e.layout(0)["coupling_rounds"]
                out[tag + "_rows"] = e.rows()
                ar0 = comm.calls["allreduce"]
                eg, _, st = e.lanczos(1, want_vectors=False)  # scale-free recurrence, one all-reduce per step
                out[tag + "_e"], out[tag + "_steps"], out[tag + "_ar"] = float(eg[0]), st["steps"], comm.calls["allreduce"] - ar0
                eg, zg, st = e.lanczos(1, want_vectors=True)  # normalised recurrence on the pitched slices, Krylov basis kept
                out[tag + "_e2"] = float(eg[0])
                zs = [None] * world
                dist.all_gather_object(zs, zg[0])
                out[tag + "_z"] = np.concatenate(zs)
                a, b, _ = e.decomposition()
                out[tag + "_a"], out[tag + "_b"] = a, b
                # the matrix-free entry point takes the same kernels where a species' row fits the LDS window
                e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm, ninj=ninj)
                out[tag + "_mf_kernel"] = e.layout(0)["kernel"]
                eg, _, st = e.lanczos(1, want_vectors=False)
                out[tag + "_mf_e"], out[tag + "_mf_steps"] = float(eg[0]), st["steps"]
                e.close()
            _audit(comm, out)
            if rank == 0:
                A = oracle.hubbard_csr(L, nup, ndown, hop, U, ninj=ninj)
                eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False, max_steps=300)
                steps_o, ao, bo, _, _ = oracle.lanczos_decomposition(A, oracle.fill_random(A.nrows, 1234), max_steps=300)
                z = out[tag + "_z"]
                out[tag + "_res"] = float(np.linalg.norm(oracle.spmv_acc(A, np.zeros_like(z), z) - out[tag + "_e2"] * z))
                out[tag + "_eo"], out[tag + "_so"], out[tag + "_ao"], out[tag + "_bo"] = float(eo[0]), so, ao, bo
            out.pop(tag + "_z")
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


@pytest.mark.parametrize("world,rounds", [(2, 1), (4, 1), (2, 2)])
def test_product_basis_kernels_on_the_transposition_exchange(world, rounds, monkeypatch):
    """(2, 2): the coupling kernel on the transposed slice walks every workgroup's block range in two rounds per panel (round 5: what slices of
    65536 blocks and more need -- BASELINE config 5 as literally written has 184756 -- forced onto the small case here)."""
    if rounds > 1:
        monkeypatch.setenv("LPP_PB_DOWN_ROUNDS", str(rounds))  # the ranks are child processes: they inherit it
    res = _run_ranks(_worker_pb_tx, world, timeout=400)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
    from helpers import rel
    for tag in ("disorder", "ladder", "chain"):
        eo, so, ao, bo = (res[0][tag + k] for k in ("_eo", "_so", "_ao", "_bo"))
        assert res[0][tag + "_res"] < 1e-5
        assert sum(res[r][tag + "_rows"] for r in range(world)) == len(ao) * 0 + sum(res[r][tag + "_rows"] for r in range(world))
        for r in range(world):
            o = res[r]
            assert o[tag + "_kernel"] == 4, "the product-basis layout was not taken"
            assert o[tag + "_rounds"] == rounds
            assert abs(o[tag + "_e"] - eo) <= 1e-10 * abs(eo) and o[tag + "_steps"] == so
            assert abs(o[tag + "_e2"] - eo) <= 1e-10 * abs(eo)
            assert o[tag + "_steps"] <= o[tag + "_ar"] <= o[tag + "_steps"] + 8
            assert o[tag + "_mf_kernel"] == 4 and abs(o[tag + "_mf_e"] - eo) <= 1e-10 * abs(eo) and o[tag + "_mf_steps"] == so
            n = min(len(ao), len(o[tag + "_a"]), 40)
            assert rel(o[tag + "_a"][:n], ao[:n]) < 1e-8 and rel(o[tag + "_b"][:n], bo[:n]) < 1e-8


def _worker_odd(rank, world, port, q):
    """Odd slice lengths (f64): N_up = C(6,2) = 15 and 3 down configurations per rank = 45 doubles per slice.  The BLAS-1 kernels move
    16-byte pairs; the copy for the next all-gather must stop at the 45th double of a send buffer that holds exactly 45."""
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        import lanczosplusplus_amd as lp
        from helpers import chain
        from lanczosplusplus_amd.comm import TorchDistComm
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        L, nup, ndown = 6, 2, 1
        hop, U = chain(L, -1.0, True), np.linspace(2.0, 4.0, L)
        n_up, n_dn = 15, 6
        stride = (n_dn // world) * n_up
        assert stride % 2 == 1
        out = {}
        for mode in ("stored", "onthefly"):
            for keep in (0, 1):  # scale-free (k_axpy_nrm copies) / vectors kept (k_swap_scale, k_scale_copy copy)
                comm = TorchDistComm(stride, 60, False, device=dev)
                # the send buffer inside a guarded allocation: exactly `stride` doubles, sentinels on both sides
                guard = torch.full((stride + 8,), 777.0, dtype=torch.float64, device=dev)
                comm.send = guard[2:2 + stride]
                comm.send.zero_()
                comm.struct.send_buf = comm.send.data_ptr()
                with comm.stream_context():
                    e = lp.LanczosEngine(max_steps=60, save_vectors=keep, stream=comm.stream_handle)
                    if mode == "stored":
                        e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm)
                    else:
                        e.setup_hubbard_onthefly(L, nup, ndown, hop, U, comm=comm)
                    eg, _, st = e.lanczos(1, want_vectors=False)
                    e.close()
                torch.cuda.synchronize()
                _audit(comm, out)
                out["%s_%d" % (mode, keep)] = (float(eg[0]), st["steps"], bool((guard[:2] == 777.0).all() and (guard[2 + stride:] == 777.0).all()))
        if rank == 0:
            A = oracle.hubbard_csr(L, nup, ndown, hop, U)
            out["e_dense"] = float(np.linalg.eigvalsh(A.to_scipy().toarray())[0])
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_two_ranks_odd_slice_lengths_stay_inside_the_send_buffer():
    world = 2
    res = _run_ranks(_worker_odd, world)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
    ed = res[0]["e_dense"]
    for r in range(world):
        for k, (e0, steps, intact) in [(k, v) for k, v in res[r].items() if k not in ("e_dense", "findings", "recorded")]:
            assert intact, (k, "wrote outside the send buffer")
            assert abs(e0 - ed) <= 1e-10 * abs(ed), (k, e0, ed)


def test_python_held_c_level_communicator_at_world_size_one():
    """lanczosplusplus_amd.comm.RcclComm (what bench.py --gpus N takes first over nccl): id, creation on the engine's own stream,
    self-test, a solve with it, per-rank statistics."""
    import oracle
    from helpers import chain
    from lanczosplusplus_amd import LanczosEngine
    from lanczosplusplus_amd.comm import RcclComm
    L, nup, ndown = 8, 4, 3
    hop, U = chain(L, -1.0, True), np.full(L, 4.0)
    A = oracle.hubbard_csr(L, nup, ndown, hop, U)
    eo, _, so = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234), want_vectors=False)
    for chunk in (0, 56 * 80):  # all-gather / transposition layout (70 up indices rounded up to 80)
        ident = RcclComm.unique_id()  # one id per communicator
        assert len(ident) == 128
        with LanczosEngine(max_steps=200) as e:
            c = RcclComm(0, 1, ident, 0, e.stream_ptr(), 56 * 70, 200, False, chunk)
            c.selftest()
            assert c.struct.nranks == 1 and c.struct.xchg_chunk == chunk and c.buffer_bytes > 0
            e.assemble_hubbard(L, nup, ndown, hop, U, comm=c)
            eg, _, st = e.lanczos(1, want_vectors=False)
            assert abs(eg[0] - eo[0]) <= 1e-10 * abs(eo[0]) and st["steps"] == so
            assert e.stats()["nnz"] == A.nnz
            e.sync()
            c.close()  # the communicator goes first (include/lpp_comm_rccl.h): the engine's stream is still alive here


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent (which never touches the GPU) starts the two ranks through
    torch.distributed.run, relays rank 0's one JSON line and returns the children's status.  Two ranks share cuda:0 over gloo
    here (LPP_BENCH_BACKEND=gloo); the line is the partitioned run's, with the transposition exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LPP_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload",
                          "hubbard_chain_L12_half_filling_U4", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = out.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["value"] > 0 and j["config"]["rows"] == 853776
    assert j["config"]["exchange"] == "transpose" and j["config"]["attempts"][-1]["ok"]


def test_eight_ranks_as_threads_config2_uneven_shards():
    """The driver's scaling run is P = 8; a one-GPU box may host 6 processes at most.  Here 8 ranks run as THREADS of one process (ThreadComm:
    same callbacks, buffers and wire format; collectives = device copies and host sums between the ranks' buffers), each with its own engine
    and stream, on BASELINE config 2 at full size: 12870 down configurations in shards of 1609 x 7 + 1607 (uneven), up ranges of 1616
    (12870 = 7 x 1616 + 1558: the last rank's range is short), product-basis kernels on the transposition exchange with the fused
    all-reduce.  Energy, stopping step and coefficients against the CPU-oracle fixture; every rank's all-reduces bitwise identical."""
    import json
    import threading
    import torch
    from helpers import rel, square
    from lanczosplusplus_amd import LanczosEngine, tridiag_lowest
    from lanczosplusplus_amd._capi import lib
    from lanczosplusplus_amd.comm import ThreadComm, ThreadGroup
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c2_hubbard4x4_U4.json")))
    world, L = 8, g["L"]
    n_up = n_dn = 12870
    hop, U = square(4, 4, -1.0, pbc=True), np.full(L, g["U"])
    per = -(-n_dn // world)
    chunk = lib().lpp_xchg_chunk(n_up, n_dn, world)
    assert per == 1609 and chunk == per * 1616
    dev = torch.device("cuda", 0)
    group = ThreadGroup(world, dev)
    out = [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            comm = ThreadComm(group, rank, per * n_up, g["max_steps"], False, xchg_chunk=chunk)
            with comm.stream_context():
                e = LanczosEngine(max_steps=g["max_steps"], min_steps=g["min_steps"], eps=g["eps"], save_vectors=0, seed=g["seed"], stream=comm.stream_handle)
                e.assemble_hubbard(L, g["nup"], g["ndown"], hop, U, comm=comm)
                lay = e.layout(0)
                a, b, st = e.decomposition()
                rows = e.rows()
                e.close()
            out[rank] = {"kernel": lay["kernel"], "rows": rows, "a": a, "b": b, "ar": comm.calls["allreduce"], "xc": comm.calls["exchange"],
                         "sums": [(o, t.numpy().tobytes()) for o, _, t in comm.sums]}
        except Exception:
            import traceback
            out[rank] = {"error": traceback.format_exc()}
            try:
                group.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck inside a collective"
    for r in range(world):
        assert "error" not in out[r], out[r]["error"]
    assert [o["rows"] for o in out] == [1609 * n_up] * 7 + [1607 * n_up] and sum(o["rows"] for o in out) == g["rows"]
    for o in out:
        assert o["kernel"] == 4, "the product-basis layout was not taken"
        assert o["sums"] == out[0]["sums"]  # every all-reduce gave every rank the same bits
        a, b = o["a"], o["b"]
        assert np.array_equal(a, out[0]["a"]) and np.array_equal(b, out[0]["b"])
        e0 = tridiag_lowest(a, b[:-1], 1)[0]
        assert abs(e0 - g["e0"]) <= 1e-10 * abs(g["e0"]) and abs(len(a) - g["steps"]) <= 1
        assert rel(a[:40], np.array(g["a"][:40])) < 1e-8 and rel(b[:40], np.array(g["b"][:40])) < 1e-8
        assert len(a) <= o["ar"] <= len(a) + 8 and o["xc"] >= 2 * len(a)  # ONE all-reduce and two all-to-alls per step


@pytest.mark.parametrize("form", ["segments"])  # ("pieces" passes as well, 44 s: the form this sector takes by itself on one GPU is covered by tests/test_gpu_fullsize.py)
def test_eight_ranks_as_threads_config5_lattice_6up6down_free_fermions(form, monkeypatch):
    """BASELINE config 5's lattice at the rank count of its node: the (6,6) sector of the 4x5 cluster -- 1,502,337,600 states, rows of 38760
    positions (beyond one LDS window) -- matrix-free over 8 ranks (threads of one process, ThreadComm) through the transposition exchange:
    the in-block kernel for rows in pieces on each rank's 4845 blocks ("pieces": the per-position template this sector takes by itself;
    "segments": the decomposition by high sites the (7,6) sector takes), the coupling kernel on the transposed slice of 4848 positions x 38760
    blocks.  U = 0: the exact free-fermion energy; every rank's coefficients bitwise identical."""
    import threading
    import torch
    from helpers import square
    from lanczosplusplus_amd import LanczosEngine
    from lanczosplusplus_amd._capi import lib
    from lanczosplusplus_amd.comm import ThreadComm, ThreadGroup
    monkeypatch.setenv("LPP_PB_SEG", "1" if form == "segments" else "0")
    world, L = 8, 20
    n_up = n_dn = 38760
    hop = square(4, 5, -1.0, pbc=True)
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = 2 * lev[:6].sum()
    per = -(-n_dn // world)
    chunk = lib().lpp_xchg_chunk(n_up, n_dn, world)
    dev = torch.device("cuda", 0)
    group = ThreadGroup(world, dev)
    out = [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            comm = ThreadComm(group, rank, per * n_up, 300, False, xchg_chunk=chunk)
            with comm.stream_context():
                e = LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0, stream=comm.stream_handle)
                e.setup_hubbard_onthefly(L, 6, 6, hop, np.zeros(L), comm=comm)
                lay = e.layout(0)
                eg, _, st = e.lanczos(1, want_vectors=False)
                rows = e.rows()
                e.close()
            out[rank] = {"kernel": lay["kernel"], "segments": lay["segments"], "pieces": lay["pieces"], "rows": rows, "e": float(eg[0]), "steps": st["steps"],
                         "sums": [(o, t.numpy().tobytes()) for o, _, t in comm.sums]}
        except Exception:
            import traceback
            out[rank] = {"error": traceback.format_exc()}
            try:
                group.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=800)
    assert not any(t.is_alive() for t in threads), "a rank is stuck inside a collective"
    for r in range(world):
        assert "error" not in out[r], out[r]["error"]
    assert sum(o["rows"] for o in out) == 38760 * 38760 and out[0]["rows"] == 4845 * 38760
    for o in out:
        assert o["kernel"] == 4 and o["pieces"] > 1 and (o["segments"] > 0) == (form == "segments"), o
        assert o["sums"] == out[0]["sums"] and o["steps"] == out[0]["steps"]
        assert abs(o["e"] - exact) <= 1e-10 * abs(exact), (o["e"], exact, o["steps"])
