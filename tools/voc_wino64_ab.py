import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(32, 80, 512, device="cuda") * 1.5 - 4
res = {}
_lib.internal_set(b"voc_wino64_k", int(os.environ.get("K64", 11)))
for rnd in range(3):
    for w64 in (0, 1):
        _lib.internal_set(b"voc_wino64", w64)
        for _ in range(2 if rnd == 0 else 1): w = voc(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): w = voc(mel)
        torch.cuda.synchronize(); res.setdefault(w64, []).append((time.perf_counter() - t0) / 3); res[("w", w64)] = w
_lib.internal_set(b"voc_wino", 0); ref = voc(mel); _lib.internal_set(b"voc_wino", 1); _lib.internal_set(b"voc_wino64", 1)
print("pair kernels at C=64: %.2f ms; two Winograd launches: %.2f ms; max|d wav| vs all-direct: %.2e / %.2e" % (min(res[0]) * 1e3, min(res[1]) * 1e3,
      float((res[("w", 0)] - ref).abs().max()), float((res[("w", 1)] - ref).abs().max())))
