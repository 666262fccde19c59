"""ctypes wrapper around oracle/nadm_oracle_c.c (the timed CPU baseline and a second checker).
TEST INFRASTRUCTURE ONLY -- imported by tests/ and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FP = C.POINTER(C.c_float)


class _Model(C.Structure):
    _fields_ = [("M", C.c_int64), ("C", C.c_int), ("K", C.c_int), ("Hd", C.c_int)] + \
               [(n, FP) for n in ("V", "P", "g", "W1", "b1", "Wk", "bk",
                                  "mV", "vV", "mP", "vP", "mg", "vg", "mW1", "vW1", "mb1", "vb1", "mWk", "vWk", "mbk", "vbk")] + \
               [("step", C.c_int)]


def _load():
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
    except OSError:
        pass
    name = "liboracle_avx2.so" if (" avx2" in flags and " fma" in flags) else "liboracle_base.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    lib.oracle_train_step.restype = C.c_double
    lib.oracle_train_step.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_set_threads.argtypes = [C.c_int]
    return lib


class CPort:
    """Single-head model state held in numpy arrays, stepped by the C port."""

    def __init__(self, V_MC, P_MK, g, W1, b1, Wk, bk):
        self.lib = _load()
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32).copy()
        self.a = {"V": f(V_MC), "P": f(P_MK), "g": f(g), "W1": f(W1), "b1": f(b1), "Wk": f(Wk), "bk": f(bk)}
        for k in list(self.a):
            self.a["m" + k] = np.zeros_like(self.a[k])
            self.a["v" + k] = np.zeros_like(self.a[k])
        self.m = _Model()
        self.m.M, self.m.C = self.a["V"].shape
        self.m.K, self.m.Hd = self.a["P"].shape[1], self.a["W1"].shape[0]
        for k, arr in self.a.items():
            setattr(self.m, k, arr.ctypes.data_as(FP))
        self.m.step = 0

    def threads(self):
        return int(self.lib.oracle_num_threads())

    def step(self, G_u8, idx, lr, apply=True, want_grads=False, want_q=False):
        G_u8 = np.ascontiguousarray(G_u8, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        M, Cc, K, Hd = self.m.M, self.m.C, self.m.K, self.m.Hd
        grads = np.empty(M * Cc + M * K + Cc + Hd * Cc + Hd + K * Hd + K, dtype=np.float32) if want_grads else None
        Q = np.empty((len(idx), K), dtype=np.float32) if want_q else None
        loss = self.lib.oracle_train_step(C.byref(self.m), G_u8.ctypes.data, idx.ctypes.data, len(idx), lr, 1 if apply else 0,
                                          grads.ctypes.data if want_grads else None, Q.ctypes.data if want_q else None)
        out = {"loss": float(loss)}
        if want_grads:
            o = 0
            for name, n, shape in (("V", M * Cc, (M, Cc)), ("P0", M * K, (M, K)), ("g", Cc, (Cc,)), ("W1", Hd * Cc, (Hd, Cc)),
                                   ("b1", Hd, (Hd,)), ("Wk0", K * Hd, (K, Hd)), ("bk0", K, (K,))):
                out[name] = grads[o:o + n].reshape(shape)
                o += n
        if want_q:
            out["Q"] = Q
        return out
