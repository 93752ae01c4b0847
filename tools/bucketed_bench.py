#!/usr/bin/env python3
"""BASELINE.json configs[3] shape of one rank (LibriTTS model, 32 ragged utterances dealt into the 256/512/768/1024
frame buckets, T = 4, fp32): sequential groups vs one HIP stream per group, per denoiser mode (1 = persistent when a
group alone pays, 2 = persistent for every group; groups admitted side by side while they fit the chip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cmtts_amd
from cmtts_amd import _lib, host, shard
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

lib = _lib.load()
DUR, N_STEPS = 6, 4
per_bucket = int(os.environ.get("PER_BUCKET", 8))
cfg = get_config("LibriTTS")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=1, dur_frames=float(DUR), dur_spread=0.0))
rs = np.random.RandomState(4)
groups = []
for bucket in shard.FRAME_BUCKETS:
    n, Lmax = per_bucket, bucket // DUR
    ln = np.maximum((rs.uniform(0.5, 1.0, size=n) * Lmax).astype(np.int64), 1); ln[0] = Lmax
    tx = rs.randint(1, cfg.n_symbols, size=(n, Lmax)).astype(np.int64)
    tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
    g = torch.Generator(device="cpu").manual_seed(bucket)
    groups.append((torch.from_numpy(tx).cuda(), torch.from_numpy(ln).cuda(), torch.randn(n, 512, generator=g).cuda(),
                   torch.randn(N_STEPS + 1, n, 1, bucket, cfg.n_mels, generator=g).cuda(), bucket))
frames = sum(int(g[1].sum()) * DUR for g in groups)


def clock(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def sequential():
    return [host.sample_with_cond(model, o["cond_ct"], o["speaker_emb"], N_STEPS, nz)
            for tx, ln, spk, nz, b in groups
            for o in [model.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=b)]]

ref = None
for mode in (1, 2):
    lib.cmtts_set_persistent_denoiser(mode)
    d = clock(sequential)
    print(f"mode {mode}: sequential groups   {d*1e3:7.2f} ms  {frames/d:9.0f} valid mel-frames/s", flush=True)
    for ns in (2, 4):
        bs = host.BucketedSynthesizer(model, N_STEPS, n_streams=ns, persistent=None)
        out = bs.run(groups); torch.cuda.synchronize()
        if ref is None:
            ref = [m.clone() for m, _ in out]
        same = all(torch.equal(m, r) for (m, _), r in zip(out, ref))
        d = clock(lambda: bs.run(groups))
        print(f"mode {mode}: {ns} streams          {d*1e3:7.2f} ms  {frames/d:9.0f} valid mel-frames/s   bitwise == first run: {same}", flush=True)
lib.cmtts_set_persistent_denoiser(1)
