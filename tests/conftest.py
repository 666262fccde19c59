import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HOOK_LIB = os.path.join(ROOT, "neural-admixture_amd", "csrc", "libnadm_testhooks.so")


def in_hook_build(request) -> bool:
    """The test hooks (nadm_test_force_slices, nadm_test_force_generic_mlp) exist only in the TEST build of the library
    (csrc/libnadm_testhooks.so, -DNADM_TEST_HOOKS); the shipping libnadm.so, which every other test loads, has none.  A test that needs
    one starts with ``if not in_hook_build(request): return``: in the ordinary process this re-runs that one test in a child process
    with NADM_LIB pointing at the test build and asserts the child passed (-> False, nothing more to do here); in the child it is True."""
    if os.environ.get("NADM_HOOK_CHILD") == "1":
        return True
    import subprocess
    assert os.path.exists(HOOK_LIB), f"{HOOK_LIB} missing: run __graft_entry__.build()"
    env = dict(os.environ, NADM_LIB=HOOK_LIB, NADM_HOOK_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", request.node.nodeid], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, "in the test build:\n" + r.stdout[-6000:] + r.stderr[-2000:]
    return False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
