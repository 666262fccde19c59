"""The reference's one native module on the hot path, by its own names.

The reference JIT-builds ``pack2bit`` from ``src/utils_c/pack2bit.cu`` with ``torch.utils.cpp_extension.load`` on every rank
(model/train.py:122-125) and calls its two entry points (pack2bit.cu:144-147):

    pack2bit.pack2bit_cpu_to_gpu(data, packed_data)              # model/train.py:121,126 -- once, the whole matrix
    pack2bit.unpack2bit_gpu_to_gpu(x_step, unpacked_step)        # model/neural_admixture.py:377-378, 404-406 -- every batch

This module exports the same two functions over ``libnadm.so`` (``nadm_pack2bit_host`` / ``nadm_unpack2bit``, include/nadm.h), so a
maintainer can hand it to ``NeuralAdmixture(..., pack2bit, ...)`` (model/train.py:131) in place of the JIT-built module:

    from neural_admixture_amd import pack2bit        # instead of load(name="pack2bit", sources=[...pack2bit.cu])

Contract, as in the reference: the CALLER allocates both tensors (``packed_data = torch.empty((N, (M + 3) // 4), uint8, device)``,
model/train.py:121; ``unpacked_step = torch.empty((b, M), uint8, device)``, neural_admixture.py:377,405), shapes are checked and a
mismatch raises ``RuntimeError`` with the reference's messages (TORCH_CHECK, pack2bit.cu:66-76,121-130), both calls BLOCK until
the result is complete (``cudaDeviceSynchronize``, pack2bit.cu:115,141).  Differences, none visible to the caller: the matrix is
packed on the host, so 2 bits per genotype cross PCIe instead of 8 (the reference ships unpacked bytes in 1024-row chunks and
packs on the device, pack2bit.cu:79-115); the kernels run on torch's current stream instead of the legacy default stream.

The training path of this package never unpacks -- its kernels decode the 2-bit codes in registers (csrc/nadm_genotype_passes.hip);
``unpack2bit_gpu_to_gpu`` exists for the reference's own model code and for tests.

Two forms of the same binding: the torch C++ extension ``csrc/ext/_pack2bit.so`` (csrc/pack2bit_ext.cpp, built by
``__graft_entry__.build()`` with torch.utils.cpp_extension like the reference builds its module) is used when it has been built;
otherwise the same two functions over ``ctypes`` below.  Same checks, same messages, same results (tests run both).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

import torch

from ._lib import lib, check, ptr

__all__ = ["pack2bit_cpu_to_gpu", "unpack2bit_gpu_to_gpu", "extension"]


def _load_extension():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "ext", "_pack2bit.so")
    if not os.path.exists(path):
        return None
    try:
        spec = importlib.util.spec_from_file_location("_pack2bit", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                          # (libnadm.so is already mapped: _lib loaded it by its soname)
        return mod
    except ImportError:                                       # built against another torch / Python: the ctypes form serves
        return None


extension = _load_extension()                                 # the torch extension module, or None

_STAGE_BYTES = 64 << 20       # pinned staging buffer, sized by BYTES (the reference stages 1024 unpacked rows at a time, pack2bit.cu:8,78;
_stage = None                 # 8192 packed rows of a 500k-SNP matrix would pin 1 GB) and kept for the process


def _check(cond: bool, msg: str) -> None:
    if not cond:                                             # TORCH_CHECK -> RuntimeError in Python
        raise RuntimeError(msg)


def _pack2bit_cpu_to_gpu_ctypes(input_cpu: torch.Tensor, output_gpu: torch.Tensor) -> None:
    """uint8 [N, M] on the CPU -> 2-bit codes [N, ceil(M / 4)] on the GPU: SNP 4c + i in bits [2i, 2i + 1] of byte c, code = value & 3,
    tail bits 0 (pack2bit.cu:10-36,65-117).  ``output_gpu`` is caller-allocated and fully written; returns None; blocks."""
    _check(input_cpu.device.type == "cpu", "Input tensor must be on CPU")
    _check(output_gpu.device.type == "cuda", "Output tensor must be on CUDA device")
    _check(input_cpu.dim() == 2 and output_gpu.dim() == 2, "pack2bit_cpu_to_gpu expects 2-D tensors")
    _check(input_cpu.dtype == torch.uint8 and output_gpu.dtype == torch.uint8, "pack2bit_cpu_to_gpu expects uint8 tensors")
    N, M = int(input_cpu.shape[0]), int(input_cpu.shape[1])
    packed_cols = (M + 3) // 4
    _check(output_gpu.shape[0] == N, "Output tensor row dimension mismatch")
    _check(output_gpu.shape[1] == packed_cols, "Output tensor column dimension mismatch")
    _check(output_gpu.is_contiguous(), "Output tensor must be contiguous")
    if N == 0 or M == 0:
        return
    src = input_cpu.contiguous()
    global _stage
    rows = max(1, min(N, _STAGE_BYTES // packed_cols))
    if _stage is None or _stage.numel() < rows * packed_cols:
        _stage = torch.empty(rows * packed_cols, dtype=torch.uint8).pin_memory()
    stage = _stage[: rows * packed_cols].view(rows, packed_cols)
    for s in range(0, N, rows):
        e = min(N, s + rows)
        check(lib.nadm_pack2bit_host(ptr(src[s:e]), ptr(stage), e - s, M, packed_cols), "pack2bit_cpu_to_gpu")
        output_gpu[s:e].copy_(stage[: e - s], non_blocking=False)       # the staging buffer is reused: wait for the copy
    torch.cuda.synchronize(output_gpu.device)


def _unpack2bit_gpu_to_gpu_ctypes(input_gpu: torch.Tensor, output_gpu: torch.Tensor) -> None:
    """2-bit codes [b, ceil(M / 4)] -> uint8 [b, M] with out[r, 4c + i] = (in[r, c] >> 2i) & 3, both on the same GPU
    (pack2bit.cu:38-62,120-142).  ``output_gpu`` is caller-allocated; returns None; blocks."""
    _check(input_gpu.device.type == "cuda", "Input tensor must be on CUDA device")
    _check(output_gpu.device.type == "cuda", "Output tensor must be on CUDA device")
    _check(input_gpu.device == output_gpu.device, "Input and Output tensors must be on the same CUDA device")
    _check(input_gpu.dim() == 2 and output_gpu.dim() == 2, "unpack2bit_gpu_to_gpu expects 2-D tensors")
    _check(input_gpu.dtype == torch.uint8 and output_gpu.dtype == torch.uint8, "unpack2bit_gpu_to_gpu expects uint8 tensors")
    N, M = int(output_gpu.shape[0]), int(output_gpu.shape[1])
    packed_cols = (M + 3) // 4
    _check(input_gpu.shape[0] == N, "Input tensor row dimension mismatch")
    _check(input_gpu.shape[1] == packed_cols, "Input tensor column dimension mismatch based on output shape")
    _check(output_gpu.is_contiguous(), "Output tensor must be contiguous")
    if N == 0 or M == 0:
        return
    src = input_gpu.contiguous()                             # a DataLoader batch of gathered rows is contiguous already
    with torch.cuda.device(output_gpu.device):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.nadm_unpack2bit(ptr(src), ptr(output_gpu), N, M, packed_cols, st), "unpack2bit_gpu_to_gpu")
        torch.cuda.current_stream().synchronize()            # the reference blocks (pack2bit.cu:141)


pack2bit_cpu_to_gpu = extension.pack2bit_cpu_to_gpu if extension is not None else _pack2bit_cpu_to_gpu_ctypes
unpack2bit_gpu_to_gpu = extension.unpack2bit_gpu_to_gpu if extension is not None else _unpack2bit_gpu_to_gpu_ctypes
