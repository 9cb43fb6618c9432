"""2 ranks sharing cuda:0 (gloo): product-basis kernels on the transposition exchange, first Lanczos coefficients vs the oracle"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch.multiprocessing as mp

def worker(rank, world, port, q):
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LPP_PRODUCT_LAYOUT="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lanczosplusplus_amd as lp
    from helpers import square
    from lanczosplusplus_amd._capi import lib
    from lanczosplusplus_amd.comm import TorchDistComm
    from math import comb
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    L, nup, ndown = 12, 6, 4
    hop, U = square(2, 6, -1.0, True), np.full(12, 4.0)
    n_up, n_dn = comb(L, nup), comb(L, ndown)
    per = -(-n_dn // world)
    out = {}
    for name, sv in (("scalefree", 0), ("scalefree_two_allreduces", 0), ("normalised", 1)):
        if name.endswith("allreduces"):
            os.environ["LPP_FUSED_ALLREDUCE"] = "0"
        else:
            os.environ.pop("LPP_FUSED_ALLREDUCE", None)
        comm = TorchDistComm(per * n_up, 40, False, device=dev, xchg_chunk=lib().lpp_xchg_chunk(n_up, n_dn, world))
        with comm.stream_context():
            e = lp.LanczosEngine(max_steps=12, eps=0.0, save_vectors=sv, stream=comm.stream_handle)
            e.assemble_hubbard(L, nup, ndown, hop, U, comm=comm)
            a, b, _ = e.decomposition()
            out[name] = (a[:6], b[:6], e.layout(0)["kernel"])
            e.close()
    q.put((rank, out))
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, 29611, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=300) for _ in range(world))
    [p.join() for p in ps]
    import oracle
    from helpers import square
    A = oracle.hubbard_csr(12, 6, 4, square(2, 6, -1.0, True), np.full(12, 4.0))
    so, ao, bo, _, _ = oracle.lanczos_decomposition(A, oracle.fill_random(A.nrows, 1234), max_steps=12, eps=0.0)
    print("oracle a", ao[:6]); print("oracle b", bo[:6])
    for name in ("scalefree", "scalefree_two_allreduces", "normalised"):
        print(name, "kernel", res[0][name][2]); print("   a", res[0][name][0]); print("   b", res[0][name][1])
