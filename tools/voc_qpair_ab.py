#!/usr/bin/env python3
"""Round 6: the fp32 generator with the k = 3 pairs of the C = 64 / 128 stages as ONE fused F(4,3) launch (conv_xlq_pair.hip; voc_qpair = 1, the default)
against the forms it replaces (voc_qpair = 0: two conv_xlq launches at C = 128, the direct pair kernel at C = 64), interleaved on one box.
Env: VB, VT (32 x 512), VN (timed passes per arm and round)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

_lib.load()
B, T, N = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512)), int(os.environ.get("VN", 5))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
outs = {}
for rnd in range(3):
    for arm in (0, 1):
        _lib.internal_set(b"voc_qpair", arm)
        w = voc(mel)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            w = voc(mel)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / N * 1e3
        outs[arm] = w.clone()
        print(f"round {rnd} voc_qpair={arm}: {ms:.3f} ms per {B} x {T}-frame batch", flush=True)
_lib.internal_set(b"voc_qpair", 1)
print(f"max|d wav| between the arms {float((outs[0] - outs[1]).abs().max()):.3e}")
