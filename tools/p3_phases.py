#!/usr/bin/env python3
"""Phase timing of resblock_pair16x3_kernel from in-kernel cycle stamps (build: K=11 tools/p3_exp.sh -DP3_STAMP=1; run with
CMTTS_LIB=cm-tts_amd/libcmtts_hip_exp.so).  Runs the fp16x3 generator once (the last pair launch of the instance the build holds leaves
its stamps) and prints mean cycles per phase over all waves."""
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
voc.set_precision("fp16x3")
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
NW = int(os.environ.get("NW", 4))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
voc(mel); torch.cuda.synchronize()
buf = torch.zeros(B * (T * 256 // 40 + 8) * NW * 10, dtype=torch.int64, device="cuda")
raw.cmtts_p3_set_debug(C.c_void_p(buf.data_ptr()))
voc(mel); torch.cuda.synchronize()
raw.cmtts_p3_set_debug(None)
s = buf.cpu().numpy().reshape(-1, 10)[:, :9]
s = s[s[:, 0] != 0]
d = np.diff(s, axis=1).astype(np.float64)
names = ["stage x", "barrier 1", "conv1 loop", "barrier 2", "xt epilogue", "barrier 3", "conv2 loop", "y epilogue"]
tot = (s[:, 8] - s[:, 0]).mean()
print(f"waves {len(s)}; mean cycles per wave per tile {tot:.0f}")
for i, (nme, v) in enumerate(zip(names, d.mean(0))):
    q = np.percentile(d[:, i], [1, 10, 50, 90, 99])
    print(f"  {nme:14s} {v:9.0f}  {100 * v / tot:5.1f} %   p1/p10/p50/p90/p99 = " + " / ".join(f"{x:.0f}" for x in q))
span = s[:, 8].max() - s[:, 0].min()
print(f"launch span {span} cycles; tiles {len(s) // NW}; span / mean tile = {span / tot:.1f}")
