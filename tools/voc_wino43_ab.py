#!/usr/bin/env python3
"""fp32 HiFi-GAN generator: the dilation-1 ResBlock convs of the Winograd path as F(2,3) tap groups (conv_xlw_kernel) vs F(4,3) (conv_xlq_kernel,
cmtts_internal_set("voc_wino43")), against the direct form.  Same process, interleaved rounds: time per batch and the differences of the waveforms.
Env: VB, VT, ROUNDS; small shapes force the Winograd forms (voc_wino = 2)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
ok = True
for B, T in [(1, 1), (2, 7), (3, 65), (1, 700), (33, 129)]:
    mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
    out = {}
    for name, wn, w43 in (("direct", 0, 0), ("f23", 2, 0), ("f43", 2, 3)):
        _lib.internal_set(b"voc_wino", wn); _lib.internal_set(b"voc_wino43", w43)
        out[name] = voc(mel).double()
    torch.cuda.synchronize()
    d23, d43 = float((out["f23"] - out["direct"]).abs().max()), float((out["f43"] - out["direct"]).abs().max())
    fin = bool(torch.isfinite(out["f43"]).all())
    print(f"B={B} T={T}: max|d wav| vs direct: F(2,3) {d23:.2e}, F(4,3) {d43:.2e}; differs from F(2,3): {not torch.equal(out['f43'], out['f23'])}; finite {fin}", flush=True)
    ok &= fin and d43 <= 1e-5 and not torch.equal(out["f43"], out["f23"])
print("ALL OK" if ok else "FAILED", flush=True)
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
_lib.internal_set(b"voc_wino", 1)
out, times = {}, {0: [], 1: [], 2: []}
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for w43 in (0, 2, 1):
        _lib.internal_set(b"voc_wino43", w43)
        for _ in range(2 if rnd == 0 else 1):
            w = voc(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            w = voc(mel)
        torch.cuda.synchronize(); times[w43].append((time.perf_counter() - t0) / 3)
        out[w43] = w
_lib.internal_set(b"voc_wino43", 1)
d = (out[1] - out[0]).double()
fl = B * T * 614.105088e6
print(f"B={B} T={T}: F(2,3) tap groups {min(times[0])*1e3:.2f} ms ({fl/min(times[0])/1e12:.1f} TFLOP/s), F(4,3) for dilation 1 only {min(times[2])*1e3:.2f} ms, the default (+ dilation 3, + dilation 5 at C = 256 or k = 3) {min(times[1])*1e3:.2f} ms ({fl/min(times[1])/1e12:.1f} TFLOP/s of the direct form's FLOPs)")
print(f"  wav: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}, finite {bool(torch.isfinite(out[1]).all())}")
