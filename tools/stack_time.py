#!/usr/bin/env python3
"""ms per T=4 sample of B x T frames through the persistent denoiser for persist_wino = WINO (0 direct, 1 F(2,3), 2 one wave per SIMD, 3 F(4,3));
CMTTS_LIB may point at a timing-only build (tools/abl_build.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 512)
cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
_lib.internal_set(b"persist_wino", int(os.environ.get("WINO", 3)))
ts = []
for rnd in range(4):
    for _ in range(2 if rnd == 0 else 1):
        host.sample_with_cond(model, cond, None, 4, noise)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        host.sample_with_cond(model, cond, None, 4, noise)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5)
print(f"{os.environ.get('CMTTS_LIB', 'product lib')} WINO={os.environ.get('WINO', 3)}: {min(ts)*1e3:.3f} ms per T=4 sample {[round(t*1e3,3) for t in ts]}")
