"""MI355X-native Neural ADMIXTURE training engine (hot path only).

Drop-in for the reference's ``neural_admixture.model.train.train`` boundary: same signature, same
returns, hand-written gfx950 HIP kernels behind a C ABI (``include/nadm.h`` / ``csrc/libnadm.so``).
PyTorch is used for device memory, streams and ``torch.distributed`` only.
"""
from . import _lib          # noqa: F401  (fails loudly if libnadm.so is missing)
from .layout import ModelLayout   # noqa: F401
from .engine import Engine        # noqa: F401
from .model import Q_P, NeuralAdmixture   # noqa: F401
from .train import train          # noqa: F401
from . import pack2bit            # noqa: F401  (the reference's native module by its own names: pack2bit.cu:144-147)

__all__ = ["train", "Engine", "ModelLayout", "Q_P", "NeuralAdmixture", "pack2bit"]
