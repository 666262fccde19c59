"""Output writers in the reference's formats (src/utils.py:36-67, src/main.py:38-44)."""
from pathlib import Path

from typing import Optional

import numpy as np
import torch


def write_outputs(Qs, run_name: str, K, min_k, max_k, out_path, Ps=None) -> None:
    """ADMIXTURE-compatible text matrices: ``{name}.{K}.Q`` [N,K] and ``{name}.{K}.P`` [M,K],
    ``np.savetxt(..., delimiter=' ')`` (default '%.18e'), exactly as src/utils.py:54-66."""
    out_path = Path(out_path)
    out_path.mkdir(parents=True, exist_ok=True)
    ks = [K] if K is not None else list(range(min_k, max_k + 1))
    jobs = [(out_path / f"{run_name}.{k}.Q", Qs[i]) for i, k in enumerate(ks)]
    if Ps is not None:
        jobs += [(out_path / f"{run_name}.{k}.P", Ps[i]) for i, k in enumerate(ks)]
    if len(jobs) <= 2:
        for path, a in jobs:
            savetxt(path, a)
        return
    # several K: the files are independent and the native writer releases the GIL -- a .P of 600k x K lines is formatted while the
    # previous one is still being written out (heads 2..10 at 600k SNPs: 0.8 GB of text)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:
        for f in [pool.submit(savetxt, path, a) for path, a in jobs]:
            f.result()                                       # (re-raises a writer's error)


def savetxt(path, a) -> None:
    """``np.savetxt(path, a, delimiter=' ')``; float32 matrices go through the native multi-threaded writer
    (nadm_savetxt_f32, byte-identical output, ~20x faster on a 600k x K matrix)."""
    import ctypes as C
    from ._lib import lib, check
    a = np.asarray(a)
    if a.dtype != np.float32 or a.ndim != 2:
        np.savetxt(path, a, delimiter=' ')
        return
    a = np.ascontiguousarray(a)
    check(lib.nadm_savetxt_f32(str(path).encode(), C.c_void_p(a.ctypes.data), a.shape[0], a.shape[1], a.shape[1]), "savetxt")


def save_model(model, name: str, save_dir: str) -> None:
    """``{name}.pt`` = state_dict without the decoders, ``{name}_config.json`` (src/main.py:40-43)."""
    Path(save_dir).mkdir(parents=True, exist_ok=True)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith('decoders')}
    torch.save(sd, f'{save_dir}/{name}.pt')
    model.save_config(name, save_dir)


class PackedGenotypes:
    """2-bit packed, sample-major genotype matrix on the host: uint8 [N, ld] (layout of include/nadm.h).
    Accepted by ``train`` / ``NeuralAdmixture.launch_training`` in place of the uint8 [N,M] tensor, so a BED file
    never has to be expanded to one byte per genotype (8(f)-1)."""

    def __init__(self, packed: torch.Tensor, N: int, M: int, flipped: bool = False):
        self.packed, self.N, self.M, self.flipped = packed, int(N), int(M), bool(flipped)
        self.shape = (self.N, self.M)

    def unpack_rows(self, s: int, e: int) -> np.ndarray:
        """uint8 [e-s, M] (host; for the init-time PCA projection only)."""
        pk = self.packed[s:e].cpu().numpy()
        out = np.empty((pk.shape[0], pk.shape[1], 4), dtype=np.uint8)
        for i in range(4):
            out[:, :, i] = (pk >> (2 * i)) & 3
        return out.reshape(pk.shape[0], -1)[:, : self.M]


def packed_chunks(data, ld: int, chunk_rows: int = 4096):
    """Yield (start, end, packed uint8 CPU tensor [end-start, ld]) over the rows of ``data`` (uint8 [N,M] CPU tensor /
    array, or PackedGenotypes), packing on the host when needed (nadm_pack2bit_host)."""
    from ._lib import lib, check, ptr
    N = data.shape[0]
    if hasattr(data, "packed"):
        for s in range(0, N, chunk_rows):
            yield s, min(N, s + chunk_rows), data.packed[s:s + chunk_rows]
        return
    t = data if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data))
    M = t.shape[1]
    for s in range(0, N, chunk_rows):
        e = min(N, s + chunk_rows)
        out = torch.empty((e - s, ld), dtype=torch.uint8)
        src = t[s:e].contiguous()
        check(lib.nadm_pack2bit_host(ptr(src), ptr(out), e - s, M, ld), "pack2bit_host")
        yield s, e, out


def _file_to_device(path, offset: int, n_bytes: int, device: torch.device, chunk: int = 16 << 20) -> torch.Tensor:
    """The file's bytes [offset, offset + n_bytes) as a uint8 tensor in HBM: read in 16 MB pieces straight into a ring of two pinned
    buffers, each copied while the next is read.  (np.fromfile + .to(device) reads the whole file into pageable memory and then copies it
    through the driver's staging buffers at 13.5 GB/s -- 0.92 s for configs[3]'s 12.5 GB on top of the read; pinned pieces go at 57 GB/s,
    underneath the read: profiles/r05_io_timing.txt.)"""
    if n_bytes < (64 << 20):                               # (pinning the ring costs more than it saves)
        return torch.from_numpy(np.fromfile(path, dtype=np.uint8, offset=offset, count=n_bytes)).to(device)
    out = torch.empty(n_bytes, dtype=torch.uint8, device=device)
    ring = [torch.empty(min(chunk, max(n_bytes, 1)), dtype=torch.uint8).pin_memory() for _ in range(2)]
    done = [None, None]
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))  # `out` may be a recycled block whose last use is still queued on the caller's stream
    with open(path, "rb", buffering=0) as f:
        f.seek(offset)
        pos, i = 0, 0
        while pos < n_bytes:
            n = min(chunk, n_bytes - pos)
            if done[i] is not None:
                done[i].synchronize()                      # the copy that last used this buffer
            view = memoryview(ring[i].numpy())[:n]
            got = 0
            while got < n:
                r = f.readinto(view[got:])
                if not r:
                    raise IOError(f"{path}: unexpected end of file")
                got += r
            with torch.cuda.stream(stream):
                out[pos:pos + n].copy_(ring[i][:n], non_blocking=True)
                done[i] = torch.cuda.Event()
                done[i].record(stream)
            pos += n
            i ^= 1
    torch.cuda.current_stream(device).wait_stream(stream)
    stream.synchronize()                                   # (the ring is freed on return)
    return out


def read_bed_packed(path: str, device: Optional[torch.device] = None, keep_on_device: bool = False) -> PackedGenotypes:
    """PLINK .bed/.fam -> :class:`PackedGenotypes`, same conventions as the reference's reader
    (src/snp_reader.py:16-45: N = lines of .fam, magic bytes skipped, M from the file size; recode table
    [2,3,1,0]; minor-allele flip when the mean code is >= 1, :109-110).  With a GPU ``device`` the 2-bit transpose runs
    there (nadm_bed_to_packed_dev: the file bytes cross PCIe once, ~1 ms for 2504 x 600k against seconds on host
    threads); ``keep_on_device`` leaves the packed matrix in HBM (single-GPU runs), otherwise it comes back to the host."""
    import ctypes as C
    from ._lib import lib, check, ptr
    from .layout import ModelLayout
    p = Path(path)
    with open(p.with_suffix(".fam")) as fam:
        N = sum(1 for _ in fam)
    nb = (N + 3) // 4
    bed_path = p.with_suffix(".bed")
    n_bytes = bed_path.stat().st_size - 3
    assert n_bytes % nb == 0, "bim file doesn't match!"
    M = n_bytes // nb
    ld = ModelLayout.row_stride(M)
    if device is not None and device.type == "cuda":
        bed_d = _file_to_device(bed_path, 3, n_bytes, device)
        out = torch.empty((N, ld), dtype=torch.uint8, device=device)
        cnt = torch.zeros(4, dtype=torch.int64, device=device)
        flp = torch.zeros(1, dtype=torch.int32, device=device)
        check(lib.nadm_bed_to_packed_dev(ptr(bed_d), N, M, ptr(out), ld, ptr(cnt), 1, ptr(flp), torch.cuda.current_stream().cuda_stream),
              "bed_to_packed_dev")
        counts, flipped = [int(v) for v in cnt.cpu()], bool(int(flp.cpu()[0]))
        del bed_d
        if not keep_on_device:
            out = out.cpu()
    else:
        B = np.fromfile(bed_path, dtype=np.uint8, offset=3)
        out = torch.empty((N, ld), dtype=torch.uint8)
        c4 = (C.c_int64 * 4)()
        fl = C.c_int32(0)
        check(lib.nadm_bed_to_packed(C.c_void_p(B.ctypes.data), N, M, ptr(out), ld, c4, 1, C.byref(fl)), "bed_to_packed")
        counts, flipped = [int(c4[i]) for i in range(4)], bool(fl.value)
    assert sum(counts) == N * M
    # the reference's check on the decoded matrix, from the code counts before the flip (src/snp_reader.py:109):
    # int(G.min()) == 0 and int(G.max()) in (2, 3)
    if counts[0] == 0 or (counts[2] == 0 and counts[3] == 0):
        raise AssertionError("Only biallelic SNPs are supported. Please make sure multiallelic sites have been removed.")
    return PackedGenotypes(out, N, M, flipped)


def orient_minor_allele(G: np.ndarray) -> np.ndarray:
    """The reference's orientation rule (src/snp_reader.py:109-110): ``G if G.mean() < 1 else 2 - G``.  There the
    subtraction is done in uint8, which turns a missing call 3 into 255.  Its GPU training path masks the code with 3 again
    when it packs (pack2bit.cu:29), i.e. missing stays missing -- that is what is written out here.  DELIBERATE DIVERGENCE on
    flipped inputs that contain missing calls: the reference's RSVD, its GMM projection and its log-likelihood consume the
    255 unmasked (and its CPU training path fails in BCE on it); here a missing call stays code 3
    everywhere (1.5 in the projections, skipped by the log-likelihood), so V, P_init and the reported log-likelihood of such
    an input differ from the reference's.  Inputs that are not flipped (mean code < 1, the usual minor-allele coding) agree."""
    assert int(G.min()) == 0 and int(G.max()) in (2, 3), \
        "Only biallelic SNPs are supported. Please make sure multiallelic sites have been removed."
    if G.mean() < 1:
        return G
    return np.where(G == 3, 3, 2 - G).astype(np.uint8)


def read_vcf(path: str) -> np.ndarray:
    """.vcf / .vcf.gz -> uint8 [n_samples, n_variants] with the reference reader's conventions (src/snp_reader.py:73-87,
    108-110): per call the two allele indices are summed, a missing allele counts -1 and a negative sum becomes 3 -- which
    makes ./1 and a haploid 1 come out as 0, exactly as scikit-allel + the reference's sum do.  The GT fields are parsed
    by nadm_vcf_parse_gt (host threads); scikit-allel itself is not needed."""
    import ctypes as C
    import gzip
    from ._lib import lib, check
    with (gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")) as f:
        buf = f.read()
    n, m = C.c_int64(0), C.c_int64(0)
    check(lib.nadm_vcf_parse_gt(buf, len(buf), C.byref(n), C.byref(m), None), "vcf_parse_gt")
    if n.value <= 0 or m.value <= 0:
        raise RuntimeError(f"{path}: no samples or no variant lines")
    G = np.empty((n.value, m.value), dtype=np.uint8)
    check(lib.nadm_vcf_parse_gt(buf, len(buf), C.byref(n), C.byref(m), C.c_void_p(G.ctypes.data)), "vcf_parse_gt")
    return orient_minor_allele(G)


def read_vcf_packed(path: str, device: Optional[torch.device] = None, keep_on_device: bool = False) -> PackedGenotypes:
    """read_vcf + the 2-bit packing of the rest of the pipeline (what read_bed_packed returns for a .bed)."""
    from ._lib import lib, check, ptr
    from .layout import ModelLayout
    G = read_vcf(path)
    N, M = G.shape
    ld = ModelLayout.row_stride(M)
    out = torch.zeros((N, ld), dtype=torch.uint8)
    check(lib.nadm_pack2bit_host(G.ctypes.data, ptr(out), N, M, ld), "pack2bit_host")
    if keep_on_device and device is not None and device.type == "cuda":
        out = out.to(device)
    return PackedGenotypes(out, N, M, False)
