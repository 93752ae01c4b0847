#!/usr/bin/env python3
"""attention_qb_kernel vs attention_kernel in isolation (cmtts_launch_attention on random q / k / v)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib
_lib.load()
lib = C.CDLL(_lib.LIB_PATH)
class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("out", C.c_void_p), ("lens", C.c_void_p), ("bstride", C.c_long), ("obstride", C.c_long),
                ("B", C.c_int), ("H", C.c_int), ("dh", C.c_int), ("L", C.c_int), ("ld", C.c_int), ("scale", C.c_float)]
lib.cmtts_launch_attention.restype = C.c_int
for B, L in ((1, 25), (2, 32), (3, 85)):
    ld = (L + 3) // 4 * 4
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, 768, ld, generator=g).cuda()
    lens = torch.full((B,), L, dtype=torch.int64).cuda()
    outs = []
    for qb in (0, 1):
        _lib.internal_set(b"attn_qb", qb)
        out = torch.zeros(B, 256, ld, device="cuda")
        a = AttnArgs(qkv.data_ptr(), out.data_ptr(), lens.data_ptr(), 768 * ld, 256 * ld, B, 2, 128, L, ld, 1.0 / np.sqrt(128.0))
        assert lib.cmtts_launch_attention(C.byref(a), None) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().astype(np.float64))
    _lib.internal_set(b"attn_qb", 1)
    x = qkv.cpu().numpy().astype(np.float64)
    ref = np.zeros((B, 256, ld))
    for b in range(B):
        for h in range(2):
            q, k, v = x[b, h * 128:(h + 1) * 128, :L], x[b, 256 + h * 128:256 + (h + 1) * 128, :L], x[b, 512 + h * 128:512 + (h + 1) * 128, :L]
            s = (k.T @ q) / np.sqrt(128.0)          # [key][query]
            p = np.exp(s - s.max(0)); p /= p.sum(0)
            ref[b, h * 128:(h + 1) * 128, :L] = v @ p
    d = np.abs(outs[0] - outs[1])
    print(B, L, "old vs qb", d.max(), "old vs f64", np.abs(outs[0] - ref).max(), "qb vs f64", np.abs(outs[1] - ref).max(), "n diff", (d > 0).sum(), "of", d.size)
    if d.max() > 0:
        idx = np.argwhere(d > 0)
        print("  rows (channel) with diffs:", sorted(set(idx[:, 1]))[:40], "cols:", sorted(set(idx[:, 2]))[:40])
