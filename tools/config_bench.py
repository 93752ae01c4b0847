#!/usr/bin/env python3
"""The five BASELINE.json configs on ONE MI355X (the per-GPU share of the 8-GPU ones), synthetic weights and inputs,
durations forced to 6 frames/phoneme: valid mel-frames/s and RTF = wall / audio seconds.  Prints a markdown table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cmtts_amd
from cmtts_amd import _lib, host, shard
from cmtts_amd.config import get_config, HifiGanConfig
from cmtts_amd.weights import synth_cmtts_state_dict, synth_hifigan_state_dict

lib = _lib.load()
DUR = 6
dev = "cuda:0"
voc = host.Generator(HifiGanConfig(), dev).load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
models = {}


def model_for(variant):
    if variant not in models:
        cfg = get_config(variant)
        models[variant] = (cfg, host.CMTotalTTS(cfg, dev).load_state_dict(
            synth_cmtts_state_dict(cfg, seed=0, dur_frames=float(DUR), dur_spread=0.0)))
    return models[variant]


def batch(cfg, B, L, seed, ragged=False):
    rs = np.random.RandomState(seed)
    ln = np.full(B, L, np.int64)
    if ragged:
        ln = np.maximum((rs.uniform(0.5, 1.0, size=B) * L).astype(np.int64), 1); ln[0] = L
    tx = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    tx[np.arange(L)[None, :] >= ln[:, None]] = 0
    spk = torch.randn(B, cfg.external_speaker_dim, device=dev) if cfg.multi_speaker else None
    return torch.from_numpy(tx).to(dev), torch.from_numpy(ln).to(dev), spk, int(ln.sum()) * DUR


def clock(fn, n, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def to_pcm(mel):
    w = voc(host.transpose_last2(mel)).squeeze(1)
    pcm = torch.empty(w.shape, dtype=torch.int16, device=w.device)
    _lib.check(lib.cmtts_wav_to_int16(host._ptr(w), host._ptr(pcm), w.numel(), 32768.0, host._stream()))
    return pcm


rows = []


def report(name, frames, d, note):
    audio = frames * 256 / 22050
    rows.append(f"| {name} | {frames} | {d*1e3:.2f} | {frames/d:,.0f} | {d/audio:.6f} | {note} |")
    print(rows[-1], flush=True)


# configs[0]: LJSpeech single utterance, T=1, batch 1 (the reference's CPU-runnable case), here on the GPU, to int16 wav
cfg, m = model_for("LJSpeech")
tx, ln, spk, fr = batch(cfg, 1, 25, 1)
nz = torch.randn(5, 1, 1, 25 * DUR, cfg.n_mels, device=dev)
def c0():
    o = m.duration_pitch_energy_net(None, tx, ln, max_mel_len=25 * DUR)
    return to_pcm(host.sample_with_cond(m, o["cond_ct"], None, 1, nz))
voc.set_precision("fp32"); m.set_precision("fp32")
report("configs[0] LJSpeech B=1, 25 phonemes, T=1, fp32, text -> int16 wav", fr, clock(c0, 20), "latency of one request")

# configs[1]: the headline (bench.py): LJSpeech B=32, 80x512, T=4, fp32, text -> mel
tx, ln, spk, fr = batch(cfg, 32, 85, 2)
nz = torch.randn(5, 32, 1, 512, cfg.n_mels, device=dev)
def c1(n=4):
    o = m.duration_pitch_energy_net(None, tx, ln, max_mel_len=512)
    return host.sample_with_cond(m, o["cond_ct"], None, n, nz)
for n in (1, 2, 4):
    report(f"configs[1] LJSpeech B=32, 80x512, T={n}, fp32, text -> mel", fr, clock(lambda: c1(n), 10), "bench.py headline" if n == 4 else "")
report("configs[1] + HiFi-GAN fp32, T=4, text -> int16 wav", fr, clock(lambda: to_pcm(c1()), 3, 1), "")
m.set_precision("fp16x3"); voc.set_precision("fp16x3")
report("configs[1] with fp16x3 operands everywhere (fp32-class: hi + lo fp16 pairs, DESIGN 3.4), T=4, text -> int16 wav", fr, clock(lambda: to_pcm(c1()), 5, 1), "")
m.set_precision("fp32"); voc.set_precision("fp32")

# configs[2]: VCTK multi-speaker B=64, T=2, bf16 + universal vocoder (bf16 ResBlock convs), 80x512
cfg, m = model_for("VCTK")
tx, ln, spk, fr = batch(cfg, 64, 85, 3)
nz = torch.randn(3, 64, 1, 512, cfg.n_mels, device=dev)
def c2(wav=True):
    o = m.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=512)
    mel = host.sample_with_cond(m, o["cond_ct"], o["speaker_emb"], 2, nz)
    return to_pcm(mel) if wav else mel
m.set_precision("bf16"); voc.set_precision("bf16")
report("configs[2] VCTK B=64, 80x512, T=2, bf16 residual blocks, text -> mel", fr, clock(lambda: c2(False), 10), "")
report("configs[2] + universal HiFi-GAN (bf16 ResBlock convs), text -> int16 wav", fr, clock(c2, 3, 1), "")
m.set_option("text16", 1)
report("configs[2] with the opt-in 16-bit text side (set_option text16: FFT-block projections / FFN and predictor convs), text -> mel", fr, clock(lambda: c2(False), 10), "durations / lengths then depend on the precision mode")
report("configs[2] text16 + universal HiFi-GAN (bf16), text -> int16 wav", fr, clock(c2, 3, 1), "")
m.set_option("text16", 0)
m.set_precision("fp32"); voc.set_precision("fp32")

# configs[3]: LibriTTS B=256 over 8 GPUs = 32 ragged utterances per rank in the 256/512/768/1024 buckets, T=4, fp32
cfg, m = model_for("LibriTTS")
groups, fr = [], 0
for bucket in shard.FRAME_BUCKETS:
    tx, ln, spk, f = batch(cfg, 8, bucket // DUR, bucket, ragged=True)
    groups.append((tx, ln, spk, torch.randn(5, 8, 1, bucket, cfg.n_mels, device=dev), bucket)); fr += f
bs = host.BucketedSynthesizer(m, 4, n_streams=4)
report("configs[3] LibriTTS, one rank's shard: 4 buckets x 8 ragged utterances, T=4, fp32, text -> mel", fr, clock(lambda: bs.run(groups), 6),
       "all bucket groups in one persistent launch per evaluation (DESIGN 3.2, RAGGED); + one RCCL all-gather per batch across ranks")

# configs[4]: zero-shot Lib->VCTK B=128 over 8 GPUs = 16 utterances per rank, 80x1024, T=4, fp16 denoiser + fp32 vocoder
tx, ln, spk, fr = batch(cfg, 16, 170, 5)
nz = torch.randn(5, 16, 1, 1024, cfg.n_mels, device=dev)
def c4(wav=True):
    o = m.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=1024)
    mel = host.sample_with_cond(m, o["cond_ct"], o["speaker_emb"], 4, nz)
    return to_pcm(mel) if wav else mel
m.set_precision("fp16")
report("configs[4] LibriTTS (zero-shot speaker vectors), one rank: B=16, 80x1024, T=4, fp16 residual blocks, text -> mel", fr, clock(lambda: c4(False), 10), "")
m.set_option("text16", 1)
report("configs[4] with text16, text -> mel", fr, clock(lambda: c4(False), 10), "")
m.set_option("text16", 0)
report("configs[4] + HiFi-GAN fp32, text -> int16 wav (end-to-end wav throughput)", fr, clock(c4, 3, 1), "")
voc.set_precision("fp16x3")
report("configs[4] + HiFi-GAN fp16x3 (fp32-class), text -> int16 wav", fr, clock(c4, 5, 1), "")
voc.set_precision("fp32")
m.set_precision("fp32")

print("\n| config (per GPU) | valid mel frames / pass | ms / pass | mel-frames/s | RTF | note |\n|---|---:|---:|---:|---:|---|")
print("\n".join(rows))
