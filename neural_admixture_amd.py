"""Import shim: the package directory is named ``neural-admixture_amd`` (not a valid Python
identifier), so ``import neural_admixture_amd`` loads it from that directory under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-admixture_amd")
_name = "neural_admixture_amd"                       # also when this file is run as ``python -m neural_admixture_amd``
_spec = importlib.util.spec_from_file_location(_name, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[_name] = _mod
_spec.loader.exec_module(_mod)
if __name__ == "__main__":                           # ``python -m neural_admixture_amd train|infer ...`` from the repo root
    from neural_admixture_amd.cli import main
    sys.exit(main())
