#!/usr/bin/env python3
"""Direct calls of cmtts_launch_conv_xl16 (conv_xl16.hip) on synthetic buffers, one (C, k, io) per run: debugging aid."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)

class XlArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long)]

Cc, k, io, dil = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B, T = 2, 320
ld = T
x = torch.randn(B, Cc, ld, device="cuda")
y = torch.zeros(B, Cc, ld, device="cuda")
res = torch.randn(B, Cc, ld, device="cuda")
wf = torch.zeros(k * Cc * Cc, dtype=torch.int16, device="cuda")
bias = torch.randn(Cc, device="cuda")
a = XlArgs(x.data_ptr(), y.data_ptr(), wf.data_ptr(), bias.data_ptr(), res.data_ptr() if io == 2 else None, Cc * ld, B, Cc, T, ld, k, dil, 0, 0.1, 0, 0, 0)
lib.cmtts_launch_conv_xl16.restype = C.c_int
rc = lib.cmtts_launch_conv_xl16(C.byref(a), 1, io, None)
torch.cuda.synchronize()
print(f"C={Cc} k={k} io={io} dil={dil}: rc={rc} finite={bool(torch.isfinite(y).all())}", flush=True)
