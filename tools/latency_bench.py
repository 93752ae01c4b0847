#!/usr/bin/env python3
"""Small-batch latency (BASELINE.json configs[0] shape on the GPU: one LJSpeech utterance, ~25 phonemes -> ~150 frames):
text->mel at T = 1 / 4 and mel->wav, per denoiser mode.  Wall time per request, stream idle between requests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config, HifiGanConfig
from cmtts_amd.weights import synth_cmtts_state_dict, synth_hifigan_state_dict

lib = _lib.load()
cfg = get_config("LJSpeech")
DUR = 6
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=float(DUR), dur_spread=0.0))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))


def clock(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


MODES = [("3-launch blocks", dict(fused=0, persist=0)), ("fused per layer", dict(fused=1, persist=0)),
         ("persistent (forced)", dict(fused=1, persist=2)), ("default", dict(fused=1, persist=1))]
for B, L in [(1, 25), (1, 85), (2, 25), (4, 25), (8, 25), (8, 85)]:
    rs = np.random.RandomState(B * 100 + L)
    texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
    T = L * DUR
    noise = torch.randn(5, B, 1, T, cfg.n_mels, device="cuda")
    audio_s = B * T * cfg.hop_length / cfg.sampling_rate
    state = {}

    def text2mel(n_steps):
        out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
        state["mel"] = host.sample_with_cond(model, out["cond_ct"], None, n_steps, noise[:n_steps + 1])

    def front():
        model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)

    print(f"B={B} L={L} T={T} ({audio_s:.2f} s of audio)", flush=True)
    print(f"   text side alone: {clock(front):.3f} ms")
    for name, m in MODES:
        lib.cmtts_set_fused_resblock(m["fused"])
        lib.cmtts_set_persistent_denoiser(m["persist"])
        r = [clock(lambda: text2mel(n)) for n in (1, 4)]
        print(f"   {name:22s} text->mel T=1 {r[0]:.3f} ms (RTF {r[0]/1e3/audio_s:.5f})   T=4 {r[1]:.3f} ms (RTF {r[1]/1e3/audio_s:.5f})", flush=True)
    lib.cmtts_set_fused_resblock(1)
    lib.cmtts_set_persistent_denoiser(1)
    for n in (1, 4):       # whole request: phoneme ids -> int16 wav (fp32 vocoder)
        def request():
            out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
            mel = host.sample_with_cond(model, out["cond_ct"], None, n, noise[:n + 1])
            w = voc(host.transpose_last2(mel)).squeeze(1)
            pcm = torch.empty(w.shape, dtype=torch.int16, device=w.device)
            _lib.check(lib.cmtts_wav_to_int16(host._ptr(w), host._ptr(pcm), w.numel(), 32768.0, host._stream()))
        e = clock(request)
        print(f"   T={n} text->int16 wav (fp32 vocoder) {e:.3f} ms (RTF {e/1e3/audio_s:.5f})", flush=True)
    text2mel(1)
    mel_ct = state["mel"].transpose(1, 2).contiguous()
    for prec in ("fp32", "bf16"):
        voc.set_precision(prec)
        v = clock(lambda: voc(mel_ct))
        print(f"   vocoder {prec}: {v:.3f} ms (RTF {v/1e3/audio_s:.5f})", flush=True)
    voc.set_precision("fp32")
