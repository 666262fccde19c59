#!/usr/bin/env python3
"""Throughput of the rows of SURVEY.md section 8 that sit either side of the training step, at configs[3]'s size, against what bounds
each (HBM ~5 TB/s achievable of 8 peak; PCIe; host memory):

  a1   pack2bit on the host + H2D of the packed bytes; pack2bit / unpack2bit on the device       (pack2bit.cu:65-147, src/loaders.py)
  f-1  PLINK .bed (SNP-major, 4 samples per byte) -> sample-major packed rows, host threads and on the device   (src/snp_reader.py:16-45)
  a15 / f-4a  the final-Q pass over all N rows (encoder only, batches of 1024 like the reference)  (neural_admixture.py:369-383)
  b-2  the .Q / .P text writers                                                                   (src/utils.py:54-66)

    python tools/io_timing.py [N M]        default 100000 500000  -> stdout (profiles/r06_io_timing.txt)"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_admixture_amd as na                                   # noqa: E402
from neural_admixture_amd._lib import lib, check, ptr                # noqa: E402
from neural_admixture_amd.layout import ModelLayout                  # noqa: E402
from neural_admixture_amd.io import savetxt                          # noqa: E402
from neural_admixture_amd.model import init_encoder_weights          # noqa: E402

N, M = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (100_000, 500_000)
dev = torch.device("cuda:0")
ld = ModelLayout.row_stride(M)


def gpu_time(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e-3)
    return best


def host_time(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    return best


print(f"# N = {N} samples x M = {M} SNPs; packed row = {ld} bytes; host threads {os.cpu_count()}")
g = torch.Generator(device=dev).manual_seed(3)

# ---- a1: pack / unpack -------------------------------------------------------------------------------------------------------------
nh = min(N, 8192)                                                  # host side: a bounded row sample (the rate is per row)
Gh = torch.randint(0, 3, (nh, M), dtype=torch.uint8)
outh = torch.empty((nh, ld), dtype=torch.uint8).pin_memory()
t = host_time(lambda: check(lib.nadm_pack2bit_host(ptr(Gh), ptr(outh), nh, M, ld), "pack_host"))
print(f"a1  pack2bit on the host ({nh} rows):              {nh * M / t / 1e9:8.2f} G genotypes/s   {nh * M / t / 1e9:6.2f} GB/s of uint8 read   ({t * N / nh:.2f} s for all {N} rows)")
outd = torch.empty((nh, ld), dtype=torch.uint8, device=dev)
t = gpu_time(lambda: outd.copy_(outh, non_blocking=True))
print(f"a1  H2D of the packed rows (pinned):              {nh * ld / t / 1e9:8.2f} GB/s                      ({t * N / nh:.3f} s for all rows; the uint8 matrix would be 4 x that)")
nd = min(N, 16384)
Gd = torch.randint(0, 4, (nd, M), dtype=torch.uint8, device=dev, generator=g)
pk = torch.empty((nd, ld), dtype=torch.uint8, device=dev)
t = gpu_time(lambda: check(lib.nadm_pack2bit(ptr(Gd), ptr(pk), nd, M, ld, None), "pack"))
print(f"a1  pack2bit on the device ({nd} rows):            {nd * M / t / 1e9:8.1f} G genotypes/s   {(nd * M + nd * ld) / t / 1e12:6.2f} TB/s moved (1 B read + 1/4 B written per genotype)")
t = gpu_time(lambda: check(lib.nadm_unpack2bit(ptr(pk), ptr(Gd), nd, M, ld, None), "unpack"))
print(f"a2  unpack2bit on the device (interop only):       {nd * M / t / 1e9:8.1f} G genotypes/s   {(nd * M + nd * ld) / t / 1e12:6.2f} TB/s moved")
del Gd, pk, Gh, outh, outd

# ---- f-1: .bed -> packed -----------------------------------------------------------------------------------------------------------
nb = (N + 3) // 4
bed_d = torch.randint(0, 256, (M, nb), dtype=torch.uint8, device=dev, generator=g)
bed_d &= ~(bed_d & ~(bed_d >> 1) & 0x55)                            # no missing calls (PLINK field 0b01 -> 0b00): like a QC'd panel
xp = torch.empty((N, ld), dtype=torch.uint8, device=dev)
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
flp = torch.zeros(1, dtype=torch.int32, device=dev)
t = gpu_time(lambda: check(lib.nadm_bed_to_packed_dev(ptr(bed_d), N, M, ptr(xp), ld, ptr(cnt), 1, ptr(flp), None), "bed_dev"))
print(f"f-1 .bed -> packed on the device (all rows):        {N * M / t / 1e9:8.1f} G genotypes/s   {(M * nb + N * ld) / t / 1e12:6.2f} TB/s moved (2 x 1/4 B per genotype)   {t * 1e3:.1f} ms (the transpose + the allele-flip pass this input asks for: {int(flp.item())})")
mh = min(M, 40_000)
bed_h = bed_d[:mh].cpu().numpy()
ldh = ModelLayout.row_stride(mh)
out_h = torch.empty((N, ldh), dtype=torch.uint8)
c4, fl = (C.c_int64 * 4)(), C.c_int32(0)
t = host_time(lambda: check(lib.nadm_bed_to_packed(C.c_void_p(bed_h.ctypes.data), N, mh, ptr(out_h), ldh, c4, 1, C.byref(fl)), "bed_host"))
print(f"f-1 .bed -> packed on host threads ({mh} SNPs):     {N * mh / t / 1e9:8.2f} G genotypes/s                         ({t * M / mh:.2f} s for all {M} SNPs)")
t0 = time.perf_counter()
bed_all = bed_d.cpu()
t_h = time.perf_counter() - t0
print(f"f-1 (the file's bytes over PCIe, pageable):         {M * nb / t_h / 1e9:8.2f} GB/s                      ({t_h:.2f} s)")
del bed_d, bed_all, bed_h, out_h

# ---- f-1 from a file: read_bed_packed (file -> pinned ring -> HBM -> transpose) against whole-file read + pageable copy -------------------
from neural_admixture_amd.io import read_bed_packed                  # noqa: E402
for nf, mf in ((2504, 600_000), (N, min(M, 100_000))):
    with tempfile.TemporaryDirectory() as td:
        nbf = (nf + 3) // 4
        raw = np.random.default_rng(5).integers(0, 256, size=mf * nbf, dtype=np.uint8)
        raw &= ~(raw & ~(raw >> 1) & 0x55)
        with open(os.path.join(td, "x.bed"), "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            raw.tofile(f)
        with open(os.path.join(td, "x.fam"), "w") as f:
            f.write("".join(f"f{i} i{i} 0 0 0 -9\n" for i in range(nf)))
        del raw

        def old():
            B = np.fromfile(os.path.join(td, "x.bed"), dtype=np.uint8, offset=3)
            return torch.from_numpy(B).to(dev)
        t_old = host_time(lambda: (old(), torch.cuda.synchronize()))
        t_new = host_time(lambda: (read_bed_packed(os.path.join(td, "x.bed"), dev, keep_on_device=True), torch.cuda.synchronize()))
        print(f"f-1 {nf} x {mf} .bed file ({mf * nbf / 1e6:.0f} MB, page cache) -> packed rows in HBM: {t_new:.3f} s "
              f"({mf * nbf / t_new / 1e9:.2f} GB/s of file);  whole-file read + pageable copy alone: {t_old:.3f} s")

# ---- a15 / f-4a: the final-Q pass --------------------------------------------------------------------------------------------------
K = 8
eng = na.Engine(M, 8, 1024, [K], dev, 1024)
eng.set_packed(xp)
rng = np.random.default_rng(1)
eng.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(0.01, 0.99, size=(K, M)).astype(np.float32),
                init_encoder_weights(42, 8, 1024, [K]))
seq = torch.arange(N, dtype=torch.int32, device=dev)


def final_q(bb):
    for s in range(0, N, bb):
        n = min(bb, N - s)
        eng.infer_q(seq[s:s + n], n)


for bb in (1024,):
    t = gpu_time(lambda: final_q(bb), reps=2)
    print(f"a15 final Q, all rows in batches of {bb}:           {N * M / t / 1e9:8.1f} G genotypes/s   {N * ld / t / 1e12:6.2f} TB/s of packed rows (sequential rows: one pass over the matrix)   {t * 1e3:.1f} ms")

# ---- b-2: writers ------------------------------------------------------------------------------------------------------------------
Q = rng.dirichlet(np.ones(K), size=N).astype(np.float32)
P = rng.uniform(0, 1, size=(M, K)).astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    t = host_time(lambda: savetxt(os.path.join(td, "a.Q"), Q))
    sz = os.path.getsize(os.path.join(td, "a.Q"))
    print(f"b-2 .Q writer [{N} x {K}]:                       {sz / t / 1e9:8.2f} GB/s of text   {t:.3f} s ({sz / 1e6:.0f} MB)")
    t = host_time(lambda: savetxt(os.path.join(td, "a.P"), P))
    sz = os.path.getsize(os.path.join(td, "a.P"))
    print(f"b-2 .P writer [{M} x {K}]:                       {sz / t / 1e9:8.2f} GB/s of text   {t:.3f} s ({sz / 1e6:.0f} MB)")
    t = host_time(lambda: np.savetxt(os.path.join(td, "b.Q"), Q[:20000], delimiter=" "), reps=1)
    print(f"b-2 (numpy.savetxt, the reference's call, 20000 rows of Q: {t:.3f} s = {t * N / 20000:.2f} s for all)")

# ---- a1 through the boundary: Engine.pack_from_host (pack on host threads, two pinned buffers, copies on a stream of their own) --------
nb_rows = min(N, 32768)
Gh = torch.randint(0, 3, (nb_rows, M), dtype=torch.uint8)
eng.pack_from_host(Gh[:1024])
torch.cuda.synchronize()
t = host_time(lambda: (eng.pack_from_host(Gh), torch.cuda.synchronize()))
print(f"a1  Engine.pack_from_host, {nb_rows} rows of uint8 from the host: {t:.3f} s = {nb_rows * M / t / 1e9:.1f} G genotypes/s ({t * N / nb_rows:.2f} s for all {N} rows)")
