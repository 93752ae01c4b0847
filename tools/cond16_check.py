#!/usr/bin/env python3
"""Direct check of cmtts_launch_cond_gemm16 (csrc/cond_gemm16.hip) against torch: fragments built here like to_fragment16."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib

class Args(C.Structure):
    _fields_ = [("X", C.c_void_p), ("Wf", C.c_void_p), ("bias", C.c_void_p), ("Y", C.c_void_p),
                ("B", C.c_int), ("T", C.c_int), ("M", C.c_int), ("K", C.c_int), ("force", C.c_int)]

raw = C.CDLL(_lib.LIB_PATH)
B, T, M, K = int(os.environ.get("CB", 2)), int(os.environ.get("CT", 200)), 512 * int(os.environ.get("CM", 10)), 256
mode = int(os.environ.get("MODE", 1))
tdt = torch.bfloat16 if mode == 1 else torch.float16
g = torch.Generator().manual_seed(1)
W = torch.randn(M, K, generator=g) * 0.05
X = torch.randn(B, K, T, generator=g)
bias = torch.randn(M, generator=g)
Wq = W.to(tdt)
# fragments [K/16][M/32][64][8]: element (k = 16 g + 8 (lane >> 5) + j, m = 32 mt + (lane & 31))
lane = np.arange(64)
kidx = (16 * np.arange(K // 16)[:, None, None, None] + 8 * (lane >> 5)[None, None, :, None] + np.arange(8)[None, None, None, :])
midx = (32 * np.arange(M // 32)[None, :, None, None] + (lane & 31)[None, None, :, None])
frag = Wq[torch.from_numpy(np.broadcast_to(midx, (K // 16, M // 32, 64, 8)).copy()), torch.from_numpy(np.broadcast_to(kidx, (K // 16, M // 32, 64, 8)).copy())].contiguous()
fd = frag.view(torch.int16).cuda()
Xd, bd = X.cuda(), bias.cuda()
Y = torch.full((B, M, T), float("nan"), device="cuda")
a = Args(Xd.data_ptr(), None, bd.data_ptr(), Y.data_ptr(), B, T, M, K, 1)
raw.cmtts_launch_cond_gemm16.argtypes = [C.POINTER(Args), C.c_void_p, C.c_int, C.c_void_p]
rc = raw.cmtts_launch_cond_gemm16(C.byref(a), C.c_void_p(fd.data_ptr()), mode, None)
torch.cuda.synchronize()
ref = torch.einsum("mk,bkt->bmt", Wq.double(), X.to(tdt).double()) + bias.double()[None, :, None]
d = (Y.cpu().double() - ref).abs()
print("rc", rc, "finite", bool(torch.isfinite(Y).all()), "max|d|", float(d.max()), "rms", float(d.pow(2).mean().sqrt()), "ref scale", float(ref.abs().max()))
bad = (d > 1e-3).nonzero()
print("bad elements", len(bad), bad[:8].tolist())
