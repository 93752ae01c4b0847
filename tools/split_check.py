#!/usr/bin/env python3
"""Round 6: the F(4,3) persistent stack with the SPLIT output projection (product build) against the same source built with -DSPLIT_PROJ=0
(tools/abl_build.sh denoiser_persist "joint:-DSPLIT_PROJ=0" -> tools/bin/libcmtts_joint.so): mel of the bench's own step (factors gathered in-kernel) and
of a dense-cp call, saved per build (run once per library; `cmp` compares the two dumps bit for bit), and ms per T = 4 sample of both paths.
Usage: python tools/split_check.py dump <tag> | python tools/split_check.py cmp <tagA> <tagB>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "split_check")
if sys.argv[1] == "cmp":
    a, b = (np.load(os.path.join(OUT, t + ".npz")) for t in sys.argv[2:4])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
        print(f"{k}: equal {np.array_equal(a[k], b[k])} max|d| {d:.3e} finite {np.isfinite(a[k]).all()}")
    sys.exit(0)
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
os.makedirs(OUT, exist_ok=True)
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
res = {}
for B, L, T in ((32, 85, 512), (33, 86, 520), (5, 20, 65)):
    rs = np.random.RandomState(B)
    tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
    noise = torch.randn(5, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(B)).cuda()
    _lib.load().cmtts_set_persistent_denoiser(2)
    out = model.duration_pitch_energy_net(None, tx, ln, max_mel_len=T)
    mel_f = host.sample_with_cond(model, out["cond_ct"], None, 4, noise, factors=out["cond_factors"])
    cond = torch.randn(B, 256, T, generator=torch.Generator().manual_seed(7)).cuda()
    mel_d = host.sample_with_cond(model, cond, None, 4, noise)
    for wn in (1, 0):
        _lib.internal_set(b"persist_wino", wn)
        res[f"wino{wn}_{B}x{T}"] = host.sample_with_cond(model, cond, None, 2, noise).cpu().numpy()
    _lib.internal_set(b"persist_wino", 3)
    torch.cuda.synchronize()
    res[f"fact_{B}x{T}"], res[f"dense_{B}x{T}"] = mel_f.cpu().numpy(), mel_d.cpu().numpy()
    if B == 32:
        for name, fn in (("factors in-kernel", lambda: host.sample_with_cond(model, out["cond_ct"], None, 4, noise, factors=out["cond_factors"])),
                         ("dense cp", lambda: host.sample_with_cond(model, cond, None, 4, noise))):
            ts = []
            for rnd in range(5):
                fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5)
            print(f"{sys.argv[2]} {name}: {min(ts) * 1e3:.3f} ms per T = 4 sample (B = 32 x 512) {[round(t * 1e3, 3) for t in ts]}", flush=True)
np.savez(os.path.join(OUT, sys.argv[2] + ".npz"), **res)
