"""r03: MEASURE the tail of pass 2 instead of arguing about it.  Pass 2 at M = 500k is 1954 blocks of 256 SNPs
on 768 resident-block slots = 2.54 rounds.  Variants timed here (HIP events around the launches, same stream, same inputs):
  one launch  [0, M)                                   main build (4 tiles of 16 SNPs per wave)
  two launches [0, m_cut) + [m_cut, M)                 both with the main build (what the r02 data-parallel plan did)
  two launches, the tail [m_cut, M) with the 128-SNP   variant build -DNADM_BF_NTW=2 (tools/abl/ntw2.so): 836 half-blocks instead of 418 blocks
where m_cut = the SNPs covered by the full rounds.  The dQ slab of the tail has twice the rows with the variant; results are not
compared here (the bit-identity of sub-range launches has its own test), only durations."""
import ctypes as C, os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neural_admixture_amd as na
from neural_admixture_amd._lib import lib, check, ptr, AdamArgs
from neural_admixture_amd.model import init_encoder_weights

dev = torch.device("cuda:0")
M, K, b, rows = 500_000, 8, 800, 8000
lib2 = C.CDLL(os.path.join(ROOT, "tools", "abl", "ntw2.so"))
fn2 = lib2.nadm_decode_bce_images
fn2.restype = C.c_int
fn2.argtypes = lib.nadm_decode_bce_images.argtypes
eng = na.Engine(M, 8, 1024, [K], dev, b)
eng.gather_batch = True
xp = torch.empty((rows, eng.ld), dtype=torch.uint8, device=dev)
Qt = torch.distributions.Dirichlet(torch.full((K,), 0.2)).sample((rows,)).float().to(dev)
Fq = torch.rand((K, M), device=dev) * 0.5
check(lib.nadm_synth_packed(ptr(xp), rows, 0, M, eng.ld, ptr(Qt), ptr(Fq), K, 0.01, 1, None))
eng.set_packed(xp)
rng = np.random.default_rng(0)
eng.load_params((0.01 * rng.standard_normal((M, 8))).astype(np.float32), rng.uniform(5e-6, 1 - 5e-6, size=(K, M)).astype(np.float32),
                init_encoder_weights(42, 8, 1024, [K]))
idx = torch.randperm(rows)[:b].to(torch.int32).to(dev)
eng.forward(idx, b)
L = eng.lay
kp = L.kp[0]
slots = torch.cuda.get_device_properties(dev).multi_processor_count * 3
chunks = (M + 255) // 256
m_cut = (chunks // slots) * slots * 256
m_cut = m_cut // 1024 * 1024
dq = torch.zeros(2 * chunks * b * kp, dtype=torch.float32, device=dev)
ls = torch.zeros(2 * chunks + 8, dtype=torch.float32, device=dev)
xg = torch.empty((b, eng.ld), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
step = [1]

def launch(fn, m0, m1, csnps):
    c0 = m0 // csnps
    ad = AdamArgs(eng._mbig.data_ptr() + (L.p_off[0] + m0 * kp) * 4, eng._vbig.data_ptr() + (L.p_off[0] + m0 * kp) * 4, 2e-3, step[0], 1.0, 0)
    rc = fn(C.c_void_p(eng.xp.data_ptr() + m0 // 4), eng.ld, ptr(idx), b, m1 - m0, C.c_void_p(eng._big.data_ptr() + (L.p_off[0] + m0 * kp) * 4), kp,
            ptr(eng.Q), L.SP, C.c_void_p(eng.gbig.data_ptr() + (L.p_off[0] + m0 * kp) * 4), C.c_void_p(dq.data_ptr() + c0 * b * kp * 4),
            C.c_void_p(ls.data_ptr() + c0 * 4), 1, C.c_void_p(xg.data_ptr() + m0 // 4), C.byref(ad), C.c_void_p(eng.qimg.data_ptr()), st)
    assert rc == 0, rc

def timed(name, body, n=60, warm=20):
    for _ in range(warm): body(); step[0] += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); body(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3); step[0] += 1
    print(f"{name:70s} {np.median(ts):7.1f} us (min {min(ts):.1f})", flush=True)

print(f"M = {M}: {chunks} blocks of 256 SNPs on {slots} slots = {chunks / slots:.2f} rounds; full rounds cover [0, {m_cut}), tail = {chunks - m_cut // 256} blocks")
for _ in range(2):
    timed("one launch [0, M), 256-SNP blocks", lambda: launch(lib.nadm_decode_bce_images, 0, M, 256))
    timed("two launches [0, m_cut) + [m_cut, M), 256-SNP blocks", lambda: (launch(lib.nadm_decode_bce_images, 0, m_cut, 256), launch(lib.nadm_decode_bce_images, m_cut, M, 256)))
    timed("two launches, tail with 128-SNP blocks (NADM_BF_NTW=2)", lambda: (launch(lib.nadm_decode_bce_images, 0, m_cut, 256), launch(fn2, m_cut, M, 128)))
    timed("full rounds only [0, m_cut)", lambda: launch(lib.nadm_decode_bce_images, 0, m_cut, 256))
    timed("tail only [m_cut, M), 256-SNP blocks", lambda: launch(lib.nadm_decode_bce_images, m_cut, M, 256))
    timed("tail only [m_cut, M), 128-SNP blocks", lambda: launch(fn2, m_cut, M, 128))
    timed("one launch [0, M), 128-SNP blocks", lambda: launch(fn2, 0, M, 128))
