import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
import test_gpu_parity as t
from lanczosplusplus_amd import LanczosEngine
for seed in range(36):
    rng=np.random.default_rng(1000+seed); cplx=bool(seed%3==2)
    pl=seed>=12; A,B=t._random_structured_csr(rng,cplx,pl)
    kernel=3 if pl else [2,3,3][seed%3]; hint=B if (kernel==3 and (pl or rng.random()<0.8)) else 0
    with LanczosEngine(dtype="c128" if cplx else "f64", spmv_kernel=kernel) as e:
        e.set_row_block(hint); e.set_csr(A.rowptr,A.colind,A.values); l=e.layout()
    print(seed, "n",A.nrows,"nnz",A.nnz,"B",B,"hint",hint,"k",l["kernel"],"coded",l["coded"],"l16",l["local16"],"tmpl",l["block_template"],"dg",l["diagonal_codes"],"stride",l["shared_stride"],"perrow",l["per_row_entries"],"shared",l["shared_entries"])
