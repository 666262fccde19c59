import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neural_admixture_amd.train import capped_host_threads
from neural_admixture_amd import gmm
import contextlib
mode = sys.argv[1]
rng=np.random.default_rng(0)
X=rng.standard_normal((4000,8))
ctx = capped_host_threads(4) if "cap" in mode else contextlib.nullcontext()
with ctx:
    if "qr" in mode:
        Y=rng.standard_normal((2504,20)).astype(np.float32); q=np.linalg.qr(Y)[0]
    if "gmm" in mode:
        m=gmm.fit_means(X,8,42)
print(mode, "phase 1 ok", flush=True)
from neural_admixture_amd._gmm_fit import fit_means as sk
import warnings; warnings.simplefilter("ignore")
r=sk(X,8,42)
print(mode, "sklearn ok", flush=True)
from threadpoolctl import threadpool_info
for p in threadpool_info(): print(p["internal_api"], p["num_threads"], p.get("version"), p.get("filepath","")[-50:])
