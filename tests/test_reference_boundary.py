"""Runs tests/golden/check_reference_boundary.py where the reference can be imported (the build container, after the scratch
build of make_golden.py's recipe): the reference's own ``src/main.py::main`` -> ``fit_model`` on top of
``neural_admixture_amd.train`` must write the same ``.pt`` keys, ``_config.json`` and ``.Q/.P`` as on top of its own ``train``.
Skipped wherever the reference is absent (the GPU box; a container without the scratch build)."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NADM_REF", "/tmp/refbuild")
DEMO = "/root/reference/demo/data/demo_data.bed"


def test_reference_fit_model_works_unchanged_on_top_of_the_drop_in_train():
    if not (os.path.exists(DEMO) and glob.glob(os.path.join(REF, "neural_admixture", "src", "utils_c", "utils*.so"))):
        pytest.skip("no scratch build of the reference here (NADM_REF; recipe in tests/golden/make_golden.py)")
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "check_reference_boundary.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, NADM_REF=REF), cwd="/tmp")
    assert r.returncode == 0 and "boundary check passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
