#!/usr/bin/env python3
"""Does the reference's own caller work UNCHANGED on top of ``neural_admixture_amd.train``?   (build container only)

Like make_golden.py this imports the reference from a scratch build (NADM_REF, recipe in make_golden.py's docstring) -- it
cannot run on the GPU box.  It executes ``neural_admixture.src.main.main`` -> ``fit_model`` (src/main.py:82-133, 19-46) twice
on the bundled demo (K = 3, 5 epochs, seed 42, the reference's CPU path):

  A. as shipped: ``fit_model`` calls the reference's ``train`` (model/train.py:19);
  B. with the one-line swap of INTEGRATION.md section 1: ``src.main.train = neural_admixture_amd.train``.  No GPU exists in the
     build container, so the engine behind the boundary is the oracle-backed test double (tests/fake_engine.py); everything
     between the reference's call and that double -- argument handling, GMM initialisation, trainer, final-Q pass, the returned
     ``(Ps, Qs, model)``, ``state_dict`` / ``save_config`` -- is the product code.

and compares what the REFERENCE'S code then writes (src/main.py:38-44): ``{name}.pt`` (state dict minus ``decoders*``),
``{name}_config.json``, ``{name}.{K}.Q`` / ``.P`` (src/utils.py:36-67).  Keys, shapes, dtypes and the JSON must be identical;
values agree to the fp32 trajectory tolerance (the reference run is forced to true fp32 like the "hi" fixtures).

    NADM_REF=/tmp/refbuild python tests/golden/check_reference_boundary.py        # prints "boundary check passed"
"""
import argparse
import json
import os
import re
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NADM_REF", "/tmp/refbuild")
DEMO = os.environ.get("NADM_DEMO", "/root/reference/demo/data/demo_data.bed")


def run(check=print):
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from neural_admixture.src import main as ref_main                     # the reference's caller, untouched
    from neural_admixture.src import utils as ref_utils
    from neural_admixture.src.svd import RSVD
    import neural_admixture_amd as na
    from fake_engine import OracleEngine

    real_prec = torch.set_float32_matmul_precision
    torch.set_float32_matmul_precision = lambda *_a, **_k: None           # neutralise the reference's 'medium' (bf16 on this CPU)
    real_prec("highest")
    try:
        ref_utils.set_seed(42)
        data, pops, N, M = ref_utils.read_data(DEMO)
        V = RSVD(data, N, M, 8, 42)
        data_t = torch.as_tensor(data, dtype=torch.uint8)
        outs = {}
        for tag in ("ref", "amd"):
            d = tempfile.mkdtemp(prefix=f"nadm_boundary_{tag}_")
            args = argparse.Namespace(epochs=5, batch_size=800, learning_rate=2e-3, save_dir=d, hidden_size=64, name="demo", seed=42,
                                      n_components=8, k=3, min_k=None, max_k=None, threads=1)
            if tag == "amd":                                              # the swap, nothing else
                ref_main.train = na.train
                na.NeuralAdmixture.engine_cls = OracleEngine
            ref_utils.set_seed(42)
            ref_main.main(0, args, 0, data_t, V.copy(), None, time.time())
            outs[tag] = d
    finally:
        torch.set_float32_matmul_precision = real_prec

    a, b = outs["ref"], outs["amd"]
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == ["demo.3.P", "demo.3.Q", "demo.pt", "demo_config.json"], (os.listdir(a), os.listdir(b))
    sa = torch.load(os.path.join(a, "demo.pt"), map_location="cpu", weights_only=True)
    sb = torch.load(os.path.join(b, "demo.pt"), map_location="cpu", weights_only=True)
    assert list(sa.keys()) == list(sb.keys()), (list(sa.keys()), list(sb.keys()))        # same keys in the same order
    assert not any(k.startswith("decoders") for k in sb)                                 # filtered by the caller (main.py:41)
    for k in sa:
        assert sa[k].shape == sb[k].shape and sa[k].dtype == sb[k].dtype, k
        assert float((sa[k] - sb[k]).abs().max()) < 2e-4, (k, float((sa[k] - sb[k]).abs().max()))
    ja, jb = json.load(open(os.path.join(a, "demo_config.json"))), json.load(open(os.path.join(b, "demo_config.json")))
    assert ja == jb and open(os.path.join(a, "demo_config.json")).read() == open(os.path.join(b, "demo_config.json")).read()
    tok = re.compile(r"^-?\d\.\d{18}e[+-]\d{2}$")                                        # np.savetxt default '%.18e'
    for ext, tol in (("Q", 1e-4), ("P", 1e-4)):
        ta, tb = open(os.path.join(a, f"demo.3.{ext}")).read(), open(os.path.join(b, f"demo.3.{ext}")).read()
        la, lb = ta.splitlines(), tb.splitlines()
        assert len(la) == len(lb) and all(len(x.split(" ")) == 3 == len(y.split(" ")) for x, y in zip(la, lb))
        assert all(tok.match(t) for t in lb[0].split(" ") + lb[-1].split(" "))
        A, B = np.loadtxt(os.path.join(a, f"demo.3.{ext}")), np.loadtxt(os.path.join(b, f"demo.3.{ext}"))
        assert A.shape == B.shape and np.abs(A - B).max() < tol, (ext, np.abs(A - B).max())
    # the inference side of the module interface: the reference's loader rebuilds its Q_P from OUR .pt + config (src/inference.py:44-57)
    from neural_admixture.model.neural_admixture import Q_P as RefQP
    ref_model = RefQP(int(jb["hidden_size"]), int(jb["num_features"]), ks_list=jb["ks"], is_train=False, V=torch.zeros(sb["V"].shape))
    missing = ref_model.load_state_dict(sb, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("decoders") for k in missing.missing_keys), missing
    check("boundary check passed: .pt keys/shapes, _config.json and .Q/.P written by the reference's fit_model agree (ref vs swapped-in train)")
    return True


if __name__ == "__main__":
    run()
