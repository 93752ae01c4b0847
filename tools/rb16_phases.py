#!/usr/bin/env python3
"""Phase timing of resblock16_kernel from in-kernel cycle stamps (build: C=64 K=7 tools/rb16_exp.sh -DRB16_STAMP=1; run with
CMTTS_LIB=cm-tts_amd/libcmtts_hip_exp.so).  Runs the bf16 generator once (the last launch of the built instance leaves its stamps)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

lib = _lib.load()
lib.cmtts_set_option(b"branch_streams", 0)
raw = C.CDLL(_lib.LIB_PATH)
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
voc.set_precision("bf16")
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
NW = int(os.environ.get("NWAVES", 8))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
voc(mel); torch.cuda.synchronize()
n = B * (T * 256 // 200 + 8) * NW * 16
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
raw.cmtts_rb16_set_debug(C.c_void_p(buf.data_ptr()))
voc(mel); torch.cuda.synchronize()
raw.cmtts_rb16_set_debug(None)
s = buf.cpu().numpy().reshape(-1, 16)
s = s[s[:, 0] != 0]
names = ["stage x + res", "barrier"] + sum([[f"p{p} conv1 loop", f"p{p} xt epi + barrier", f"p{p} conv2 loop", f"p{p} x epi + barrier"] for p in range(3)], []) + ["y store"]
cols = [0, 1, 2] + list(range(3, 15)) + [15]
t = s[:, cols]
d = np.diff(t, axis=1).astype(np.float64)
tot = (s[:, 15] - s[:, 0]).mean()
print(f"waves {len(s)}; mean cycles per wave per tile {tot:.0f}")
for i, nme in enumerate(names):
    q = np.percentile(d[:, i], [1, 50, 99])
    print(f"  {nme:22s} {d[:, i].mean():9.0f}  {100 * d[:, i].mean() / tot:5.1f} %   p1/p50/p99 = " + " / ".join(f"{x:.0f}" for x in q))
