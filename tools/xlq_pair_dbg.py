#!/usr/bin/env python3
"""Debug: conv_xlq_pair vs two conv_xlq launches — where do they differ?"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cmtts_amd
from cmtts_amd import _lib
from test_gpu_parity import _pack_wino43
lib = C.CDLL(_lib.LIB_PATH)
DEV = "cuda:0"
class XlArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long),
                ("wino_force", C.c_int), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("row_split", C.c_int)]
class PairArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("w1f", C.c_void_p), ("b1", C.c_void_p), ("w2f", C.c_void_p), ("b2", C.c_void_p),
                ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int), ("dil", C.c_int),
                ("accum", C.c_int), ("slope", C.c_float), ("dbg", C.c_void_p)]
lib.cmtts_launch_conv_xlq.restype = C.c_int
lib.cmtts_launch_conv_xlq_pair.restype = C.c_int
for Cc, dil, T, ld in [(128, 1, 66, 68), (128, 1, 60, 60), (128, 1, 8, 8), (64, 1, 131, 131), (128, 3, 200, 203), (64, 5, 300, 300)]:
    rs = np.random.RandomState(Cc + 7 * dil + T)
    B = 2
    x = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    y0 = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    w1 = (rs.standard_normal((Cc, Cc, 3)) / np.sqrt(Cc * 3)).astype(np.float32)
    w2 = (rs.standard_normal((Cc, Cc, 3)) / np.sqrt(Cc * 3)).astype(np.float32)
    b1 = rs.standard_normal(Cc).astype(np.float32)
    b2 = rs.standard_normal(Cc).astype(np.float32)
    xd, b1d, b2d = (torch.from_numpy(v).to(DEV) for v in (x, b1, b2))
    w1f = torch.from_numpy(_pack_wino43(w1)).to(DEV)
    w2f = torch.from_numpy(_pack_wino43(w2)).to(DEV)
    for accum in (0, 1):
        xt = torch.full((B, Cc, ld), 7.0, device=DEV)
        yref = torch.from_numpy(y0).to(DEV)
        a1 = XlArgs(xd.data_ptr(), xt.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), None, Cc * ld, B, Cc, T, ld, 3, dil, 0, 0.1, 0, 0, 0, 1, None, None, 0.0, 0)
        assert lib.cmtts_launch_conv_xlq(C.byref(a1), None) == 0
        a2 = XlArgs(xt.data_ptr(), yref.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), xd.data_ptr(), Cc * ld, B, Cc, T, ld, 3, 1, accum, 0.1, 0, 0, 0, 1, None, None, 0.0, 0)
        assert lib.cmtts_launch_conv_xlq(C.byref(a2), None) == 0
        yd = torch.from_numpy(y0).to(DEV)
        pa = PairArgs(xd.data_ptr(), yd.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), Cc * ld, B, Cc, T, ld, 3, dil, accum, 0.1, None)
        assert lib.cmtts_launch_conv_xlq_pair(C.byref(pa), None) == 0
        torch.cuda.synchronize()
        got, two = yd.cpu().numpy(), yref.cpu().numpy()
        d = np.abs(got - two)
        nz = np.argwhere(d > 0)
        print(f"C={Cc} dil={dil} T={T} ld={ld} accum={accum}: max {d.max():.3e}, {len(nz)} of {d.size} differ; beyond T equal {np.array_equal(got[:, :, T:], y0[:, :, T:])}")
        if len(nz):
            cols = np.unique(nz[:, 2]); rows = np.unique(nz[:, 1])
            print("   columns", cols[:40], "... rows", rows[:20], len(rows))
