#!/usr/bin/env python3
"""Text side alone (encoder + variance adaptor + length regulator + pitch + conditioner GEMM) at a given B, L:
run under `rocprofv3 --kernel-trace` and summarise with tools/text_side_summary.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
B, L = int(os.environ.get("PB", 1)), int(os.environ.get("PL", 25))
cfg = get_config(os.environ.get("PV", "LJSpeech"))
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
rs = np.random.RandomState(0)
texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
spk = torch.randn(B, 512, device="cuda") if cfg.multi_speaker else None
noise = torch.randn(2, B, 1, L * 6, cfg.n_mels, device="cuda")
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=L * 6)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mel = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], 1, noise)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B} L={L}: text side {1e3*(t1-t0):.3f} ms, one denoiser evaluation {1e3*(t2-t1):.3f} ms")
from cmtts_amd import _lib
lib = _lib.load()
for flag in (0, 1):
    lib.cmtts_set_option(b"branch_streams", flag)
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=L * 6)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"branch_streams={flag}: text side median {1e3*sorted(ts)[len(ts)//2]:.3f} ms")
