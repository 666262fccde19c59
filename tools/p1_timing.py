#!/usr/bin/env python3
"""Pass 1 on the bf16 instruction (nadm_encode_fwd) against pass 1 on the FP4 x FP6 instruction with V as an operand image
(nadm_v_image + nadm_encode_fwd_img), b = 800 / 100 rows of 100k x 500k and 2504 x 600k: kernel times by HIP events.  -> stdout"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neural_admixture_amd as na                                  # noqa: E402
from neural_admixture_amd._lib import lib, check, ptr              # noqa: E402

dev = torch.device("cuda:0")
for rows, M, b in ((20000, 500_000, 800), (20000, 500_000, 100), (2504, 600_000, 800), (4000, 1_000_000, 800)):
    e = na.Engine(M, 8, 1024, [8], dev, b)
    L = e.lay
    xp = torch.randint(0, 256, (rows, e.ld), dtype=torch.uint8, device=dev)
    e.set_packed(xp)
    e.pflat.normal_()
    idx = torch.randperm(rows, device=dev)[:b].to(torch.int32)
    vimg = torch.empty(int(lib.nadm_v_image_bytes(M)), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    V = ptr(e.pflat[L.off_v: L.off_v + M * L.CP])

    def t(fn, n=50):
        for _ in range(10):
            fn()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        z.record()
        torch.cuda.synchronize()
        return a.elapsed_time(z) / n * 1e3
    t_bf = t(lambda: check(lib.nadm_encode_fwd(ptr(xp), e.ld, ptr(idx), b, M, V, L.CP, ptr(e.zpart), st)))
    t_im = t(lambda: check(lib.nadm_v_image(V, M, L.CP, ptr(vimg), st)))
    t_f4 = t(lambda: check(lib.nadm_encode_fwd_img(ptr(xp), e.ld, ptr(idx), b, M, ptr(vimg), L.CP, ptr(e.zpart), 0, None, 0, 0, None, None, None, st)))
    print(f"rows {rows} M {M} b {b}: bf16 pass 1 {t_bf:.1f} us | image of V {t_im:.1f} us | FP4 x FP6 pass 1 {t_f4:.1f} us")
    del e, xp, vimg
