#!/usr/bin/env python3
"""BASELINE.json configs[3] on one rank (bench.py's 4 x 8-utterance LibriTTS shard): BucketedSynthesizer modes side by side.
MODE=streams|ragged|ragged_untrimmed (default: all), N = timed passes.  The target of rocprofv3 runs (MODE=ragged N=3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, shard
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

DUR, N_STEPS, dev = 6, 4, "cuda:0"
lcfg = get_config("LibriTTS")
lmodel = host.CMTotalTTS(lcfg, dev).load_state_dict(synth_cmtts_state_dict(lcfg, seed=1, dur_frames=float(DUR), dur_spread=0.0))
rs4 = np.random.RandomState(4)
groups, valid = [], 0
for bucket in shard.FRAME_BUCKETS:
    n, Lmax = 8, bucket // DUR
    ln = np.maximum((rs4.uniform(0.5, 1.0, size=n) * Lmax).astype(np.int64), 1)
    ln[0] = Lmax
    tx = rs4.randint(1, lcfg.n_symbols, size=(n, Lmax)).astype(np.int64)
    tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
    gen4 = torch.Generator(device="cpu").manual_seed(bucket)
    groups.append((torch.from_numpy(tx).to(dev), torch.from_numpy(ln).to(dev), torch.randn(n, lcfg.external_speaker_dim, generator=gen4).to(dev),
                   torch.randn(N_STEPS + 1, n, 1, bucket, lcfg.n_mels, generator=gen4).to(dev), bucket))
    valid += int(ln.sum()) * DUR
N = int(os.environ.get("N", 10))
for mode in os.environ.get("MODE", "streams,ragged,ragged_untrimmed").split(","):
    bs = host.BucketedSynthesizer(lmodel, N_STEPS, n_streams=4, mode="streams" if mode == "streams" else "ragged", trim=mode != "ragged_untrimmed")
    for _ in range(2):
        bs.run(groups)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        out = bs.run(groups)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / N
    print(f"{mode}: {d*1e3:.2f} ms per shard, {valid/d:.0f} valid mel-frames/s ({valid} valid frames)", flush=True)
host.check_async_error()
