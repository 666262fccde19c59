"""Output writers in the reference's formats (src/utils.py:36-67, src/main.py:38-44)."""
from pathlib import Path

import numpy as np
import torch


def write_outputs(Qs, run_name: str, K, min_k, max_k, out_path, Ps=None) -> None:
    """ADMIXTURE-compatible text matrices: ``{name}.{K}.Q`` [N,K] and ``{name}.{K}.P`` [M,K],
    ``np.savetxt(..., delimiter=' ')`` (default '%.18e'), exactly as src/utils.py:54-66."""
    out_path = Path(out_path)
    out_path.mkdir(parents=True, exist_ok=True)
    ks = [K] if K is not None else list(range(min_k, max_k + 1))
    for i, k in enumerate(ks):
        np.savetxt(out_path / f"{run_name}.{k}.Q", Qs[i], delimiter=' ')
        if Ps is not None:
            np.savetxt(out_path / f"{run_name}.{k}.P", Ps[i], delimiter=' ')


def save_model(model, name: str, save_dir: str) -> None:
    """``{name}.pt`` = state_dict without the decoders, ``{name}_config.json`` (src/main.py:40-43)."""
    Path(save_dir).mkdir(parents=True, exist_ok=True)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith('decoders')}
    torch.save(sd, f'{save_dir}/{name}.pt')
    model.save_config(name, save_dir)
