#!/usr/bin/env python3
"""Time cmtts_launch_conv_xl16 (conv_xl16.hip) alone at the vocoder's shapes (B = 32, 512 frames): C = 128 -> 32768 columns, C = 256 -> 4096.
CMTTS_LIB selects an ablation build (tools/xl16_ablation.sh)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)

class XlArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long)]
lib.cmtts_launch_conv_xl16.restype = C.c_int
tag = os.path.basename(_lib.LIB_PATH)
for Cc, T in ((128, 32768), (256, 4096)):
    B = 32
    Tr, T = T, T + int(os.environ.get("LDPAD", "0"))          # LDPAD: row stride = T + pad elements (channel-conflict experiment); the convs still cover Tr columns
    x = torch.randn(B, Cc, T, device="cuda")
    x16 = torch.randn(B, Cc, T, device="cuda").to(torch.bfloat16)
    y = torch.zeros(B, Cc, T, device="cuda")
    y16 = torch.zeros(B, Cc, T, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(B, Cc, T, device="cuda")
    bias = torch.randn(Cc, device="cuda")
    for k in (3, 7, 11):
        wf = (torch.randn(k * Cc * Cc, device="cuda") * 0.02).to(torch.bfloat16)
        for io, dil in ((1, 5), (2, 1)):
            if io == 1:
                a = XlArgs(x.data_ptr(), y16.data_ptr(), wf.data_ptr(), bias.data_ptr(), None, Cc * T, B, Cc, Tr, T, k, dil, 0, 0.1, 0, 0, 0)
            else:
                a = XlArgs(x16.data_ptr(), y.data_ptr(), wf.data_ptr(), bias.data_ptr(), res.data_ptr(), Cc * T, B, Cc, Tr, T, k, dil, 1, 0.1, 0, 0, 0)
            for _ in range(3):
                rc = lib.cmtts_launch_conv_xl16(C.byref(a), 1, io, None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                lib.cmtts_launch_conv_xl16(C.byref(a), 1, io, None)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            fl = 2.0 * Cc * Cc * k * Tr * B
            print(f"{tag} C={Cc} k={k:2d} io={io}: rc={rc} {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
