"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against
(a) the committed golden vectors captured from the reference and (b) the numpy oracle on the same
seeded inputs.  Tolerances: bit-exact for integer/index work; |delta mel| < 1e-3 (north-star fp32
bound) for floating point, tighter where the stage is short."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

import cmtts_amd
from cmtts_amd import _lib
from cmtts_amd.config import get_config, HifiGanConfig
from cmtts_amd.weights import synth_cmtts_state_dict, synth_hifigan_state_dict
from oracle import cmtts_oracle as O
from conftest import golden_noise, pitch_margin_mask, pitch_flips, near_flip_mask, report, conv_form, same_result, WINO_TOL, voc_form, same_wav, same_pcm, VOC_WINO_TOL, same_trimmed, trim_exact  # noqa: F401

# test_bucketed_ragged_shard_vs_oracle (unsearched inputs: no margins): measured on MI355X.  Rounds 2-5: none.  Round 6: ONE pitch frame of 4096 — the text side's
# fp32 rounding changed (FFN conv as F(2,3) tap groups, the pitch predictor's k = 5 convs as F(4,3) tap groups, the softmax's roundings pinned) and one frame
# whose pre-rounding bucket value lies within conftest.FLIP_MARGIN of a rounding boundary went to the neighbouring bucket; the test asserts exactly that
# (on a boundary, off by one).
KNOWN_ORACLE_FLIPS = {"energy": 0, "pitch": 1}

pytestmark = pytest.mark.gpu
VARIANTS = ["LJSpeech", "VCTK", "LibriTTS"]
DEV = "cuda:0"


def _host():
    from cmtts_amd import host
    return host


@pytest.fixture(scope="module")
def models(golden):
    host = _host()
    cache = {}

    def get(variant):
        if variant not in cache:
            g = golden("cmtts_" + variant)
            cfg = get_config(variant)
            sd = synth_cmtts_state_dict(cfg, seed=int(g["seed"]), dur_frames=4.0, dur_spread=0.03)
            cache[variant] = (g, cfg, sd, host.CMTotalTTS(cfg, DEV).load_state_dict(sd))
        return cache[variant]
    return get


def _np(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- kernel level

@pytest.mark.parametrize("Cin,Cout,K,dil,T,B", [
    (256, 512, 3, 1, 300, 2),      # denoiser k3 (128x128 tile)
    (80, 256, 1, 1, 97, 3),        # input projection, ragged T, Cin not a multiple of 16? (80 = 5 chunks)
    (256, 80, 1, 1, 130, 2),       # output projection (64x256 tile, Cout not a multiple of 32)
    (64, 64, 11, 5, 700, 1),       # HiFi-GAN k11 d5 halo 50
    (32, 32, 7, 3, 1000, 2),       # 32x256 tile
    (256, 1024, 9, 1, 85, 2),      # FFN k9
    (128, 11, 5, 1, 40, 2),        # tiny Cout
    (20, 36, 3, 1, 1, 1),          # degenerate T = 1, odd channel counts
])
def test_conv1d_kernel(Cin, Cout, K, dil, T, B):
    lib = _lib.load()
    rs = np.random.RandomState(Cin * 7 + Cout + K)
    x = rs.standard_normal(size=(B, Cin, T)).astype(np.float32)
    w = (rs.standard_normal(size=(Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    w += (np.arange(Cout)[:, None, None] * 1e-3).astype(np.float32)          # asymmetric: catches transposes
    b = rs.standard_normal(size=(Cout,)).astype(np.float32)
    pad = (K - 1) * dil // 2
    ref = O.conv1d(x, w, b, padding=pad, dilation=dil)
    xd, bd = torch.from_numpy(x).to(DEV), torch.from_numpy(b).to(DEV)
    packed, ld = C.c_void_p(), C.c_int()
    _lib.check(lib.cmtts_pack_conv_weight(w.ctypes.data_as(C.c_void_p), Cout, Cin, K, C.byref(packed), C.byref(ld)))
    y = torch.full((B, Cout, ref.shape[2]), float("nan"), device=DEV)
    _lib.check(lib.cmtts_conv1d(xd.data_ptr(), packed, ld.value, bd.data_ptr(), B, Cin, Cout, T, K, dil, pad, 0,
                                y.data_ptr(), None))
    torch.cuda.synchronize()
    lib.cmtts_free_device(packed)
    np.testing.assert_allclose(_np(y), ref, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("B,L,C,maxd,T", [(3, 20, 256, 9, None), (4, 7, 32, 3, 10), (256, 171, 256, 12, 1024)])
def test_length_regulator_bit_exact(B, L, C, maxd, T):
    """LengthRegulator + dur_to_mel2ph: pure index work -> bit-exact, incl. zero durations, ragged
    lengths, truncation to T and (last case) the full cfg4 size."""
    lib = _lib.load()
    rs = np.random.RandomState(B + L)
    x = rs.standard_normal(size=(B, L, C)).astype(np.float32)
    dur = rs.randint(0, maxd + 1, size=(B, L)).astype(np.float32)
    lens = rs.randint(1, L + 1, size=(B,))
    pad = np.arange(L)[None, :] >= lens[:, None]
    dur[pad] = 0
    ref, ref_len = O.length_regulate(x, dur, T)
    Tn = ref.shape[1]
    ref_m2p = O.dur_to_mel2ph(dur, pad, Tn)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(DEV)
    dd = torch.from_numpy(dur).to(DEV)
    out = torch.empty(B, C, Tn, device=DEV)
    m2p = torch.empty(B, Tn, dtype=torch.int64, device=DEV)
    mlen = torch.empty(B, dtype=torch.int64, device=DEV)
    cum = torch.empty(B, L, dtype=torch.int32, device=DEV)
    _lib.check(lib.cmtts_length_regulate(xd.data_ptr(), dd.data_ptr(), B, C, L, Tn, out.data_ptr(), m2p.data_ptr(),
                                         mlen.data_ptr(), cum.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(_np(mlen), ref_len)
    assert np.array_equal(_np(m2p), ref_m2p)
    assert np.array_equal(_np(out).transpose(0, 2, 1), ref)            # bit-exact copy
    # size-independent property: counting frames per phoneme recovers the (truncated) durations
    m = _np(m2p)
    cnt = np.stack([np.bincount(m[b], minlength=L + 1)[1:] for b in range(B)])
    full = dur.astype(np.int64)
    assert np.array_equal(cnt.sum(1), np.minimum(full.sum(1), Tn))
    untrunc = full.sum(1) <= Tn
    assert np.array_equal(cnt[untrunc], full[untrunc])


# ----------------------------------------------------------------------------- stage level vs golden

@pytest.mark.parametrize("variant", VARIANTS)
def test_duration_pitch_speaker_net_golden(models, variant):
    g, cfg, sd, model = models(variant)
    spk = torch.from_numpy(g["spker_embeds"]) if cfg.multi_speaker else None
    out = model.duration_pitch_energy_net(None, torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]),
                                          spker_embeds=spk)
    torch.cuda.synchronize()
    L = g["texts"].shape[1]
    valid = np.arange(L)[None, :] < g["src_lens"][:, None]
    np.testing.assert_allclose(_np(out["enc_out"]), g["enc_out"], atol=5e-5)
    np.testing.assert_allclose(_np(out["log_d_predictions"]), g["log_d"], atol=5e-5)
    np.testing.assert_allclose(_np(out["e_predictions"]), g["e_pred"], atol=1e-4)
    # integer / index stages: bit-exact
    assert np.array_equal(_np(out["d_rounded"]), g["d_rounded"])
    assert np.array_equal(_np(out["mel_lens"]), g["mel_len"])
    assert np.array_equal(_np(out["mel2ph"]), g["mel2ph"])
    assert np.array_equal(_np(out["e_idx"])[valid], g["e_idx"][valid])
    assert np.array_equal(_np(out["mel_masks"]), g["mel_mask"])
    if cfg.multi_speaker:
        np.testing.assert_allclose(_np(out["speaker_emb"]), g["speaker_emb"], atol=2e-5)
    pp = out["p_predictions"]
    np.testing.assert_allclose(_np(pp["cwt"]), g["cwt_out"], atol=2e-4)
    np.testing.assert_allclose(_np(pp["f0_mean"]), g["f0_mean"], atol=2e-5)
    np.testing.assert_allclose(_np(pp["f0_std"]), g["f0_std"], atol=2e-5)
    np.testing.assert_allclose(_np(pp["f0_denorm"]), g["f0_denorm"], rtol=5e-4, atol=1e-2)
    same = pitch_flips(_np(pp["p_idx"]), g["p_idx"], g["f0_denorm"], "cmtts_" + variant)
    np.testing.assert_allclose(_np(out["cond"])[same], g["cond"][same], atol=5e-5)


def test_speaker_table_variant_golden(golden):
    """preprocess.yaml `speaker_embedder: none` (model/cmtts.py:26-38,77-78): speaker_emb = nn.Embedding(n_speaker, 256)
    indexed by `speakers` — the reference's own output for that branch (tests/golden/cmtts_VCTK_table.npz), end to end
    through synthesize()'s call surface with T = 2."""
    host = _host()
    g = golden("cmtts_VCTK_table")
    cfg = get_config("VCTK_table")
    sd = synth_cmtts_state_dict(cfg, seed=int(g["seed"]), dur_frames=4.0, dur_spread=0.03)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    speakers, texts, lens = torch.from_numpy(g["speakers"]), torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"])
    out = model.duration_pitch_energy_net(speakers, texts, lens)
    torch.cuda.synchronize()
    assert np.array_equal(_np(out["speaker_emb"]), g["speaker_emb"])              # a gather: bit-exact
    np.testing.assert_allclose(_np(out["log_d_predictions"]), g["log_d"], atol=5e-5)
    assert np.array_equal(_np(out["d_rounded"]), g["d_rounded"])
    assert np.array_equal(_np(out["mel_lens"]), g["mel_len"]) and np.array_equal(_np(out["mel2ph"]), g["mel2ph"])
    same = pitch_flips(_np(out["p_predictions"]["p_idx"]), g["p_idx"], g["f0_denorm"], "cmtts_VCTK_table")
    np.testing.assert_allclose(_np(out["cond"])[same], g["cond"][same], atol=5e-5)
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[Gen.i]).to(DEV)
            Gen.i += 1
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    batch = (["a", "b", "c"], ["x", "y", "z"], speakers, texts, lens, int(lens.max()), None)
    res = host.CMTotalTTSSynthesize.from_model(model, T=2, generator=Gen()).synthesize(batch)
    host.synchronize()
    assert np.abs(_np(res[0]) - g["mel_T2"])[near_flip_mask(same)].max() < 1e-3
    with pytest.raises(AssertionError):
        model.duration_pitch_energy_net(None, texts, lens)                         # speakers are required
    with pytest.raises(IndexError):
        model.duration_pitch_energy_net(torch.tensor([0, 1, cfg.n_speaker]), texts, lens)


@pytest.mark.parametrize("variant", VARIANTS)
def test_denoiser_forward_golden(models, variant):
    """CMDenoiserTTS.forward on the reference's own conditioning."""
    g, cfg, sd, model = models(variant)
    spk = torch.from_numpy(g["speaker_emb"]).to(DEV) if cfg.multi_speaker else None
    out = model.net(torch.from_numpy(g["den_x"]), torch.from_numpy(g["den_t"]), torch.from_numpy(g["cond"]), spk, None)
    torch.cuda.synchronize()
    assert out.shape == g["den_out"].shape
    err = np.abs(_np(out) - g["den_out"]).max()
    assert err < 1e-3, err
    assert err < 3e-4, f"suspiciously large fp32 drift {err}"


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_steps", [1, 2, 4])
def test_sampler_golden(models, variant, n_steps):
    """cmtts_sample (hoisted encoder, fused preconditioning) == reference karras_sample_tts."""
    host = _host()
    g, cfg, sd, model = models(variant)
    B, T, _ = g["cond"].shape
    noise = np.stack(golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5))
    cond_ct = torch.from_numpy(np.ascontiguousarray(g["cond"].transpose(0, 2, 1))).to(DEV)
    spk = torch.from_numpy(g["speaker_emb"]).to(DEV) if cfg.multi_speaker else None
    mel = host.sample_with_cond(model, cond_ct, spk, n_steps, torch.from_numpy(noise).to(DEV))
    torch.cuda.synchronize()
    err = np.abs(_np(mel) - g[f"mel_T{n_steps}"]).max()
    assert err < 1e-3, err


@pytest.mark.parametrize("variant", ["LJSpeech", "VCTK"])
def test_karras_sample_tts_end_to_end(models, variant):
    """The reference-shaped call: phoneme ids in, mel out (synthesize.py:111-147), T = 4."""
    host = _host()
    g, cfg, sd, model = models(variant)
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        def __init__(self):
            self.i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[self.i]).to(DEV)
            self.i += 1
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    kw = dict(speakers=None, texts=torch.from_numpy(g["texts"]), src_lens=torch.from_numpy(g["src_lens"]),
              spker_embeds=torch.from_numpy(g["spker_embeds"]) if cfg.multi_speaker else None)
    diffusion = host.KarrasDenoiser(distillation=True)
    mel = host.karras_sample_tts(diffusion, model, (B, 1, T, cfg.n_mels), steps=2, model_kwargs=kw, device=DEV,
                                 sampler="multistep", ts=(0, 0, 0, 0, 1), generator=Gen())
    torch.cuda.synchronize()
    ref = g["mel_T4"]
    # frames whose pitch bucket sits on a rounding boundary may take the neighbouring embedding row;
    # the denoiser's receptive field (+-20 frames) spreads that, so compare where all buckets agree
    ok = _mask_near_pitch_flips(model, g, kw, "cmtts_" + variant)
    err = np.abs(_np(mel) - ref)
    assert err[ok].max() < 1e-3, err[ok].max()


def _mask_near_pitch_flips(model, g, kw, tag):
    """Frames of the end-to-end mel that can be compared with the golden: all of them when no pitch bucket flipped
    (the pinned count, conftest.KNOWN_PITCH_FLIPS); otherwise those outside the denoiser's reach of a flipped frame."""
    out = model.duration_pitch_energy_net(None, kw["texts"], kw["src_lens"], spker_embeds=kw["spker_embeds"])
    same = pitch_flips(_np(out["p_predictions"]["p_idx"]), g["p_idx"], g["f0_denorm"], tag)
    return near_flip_mask(same)


@pytest.mark.parametrize("T_steps", [1, 2, 4])
def test_synthesize_driver(models, T_steps):
    """CMTotalTTSSynthesize.synthesize (synthesize.py:88-153): the 7-tuple batch in, out_put[0/10/11] out."""
    host = _host()
    g, cfg, sd, model = models("VCTK")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[Gen.i]).to(DEV)
            Gen.i += 1
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    texts, lens = torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"])
    spk = torch.from_numpy(g["spker_embeds"])
    batch = (["a", "b", "c"], ["x", "y", "z"], torch.zeros(B, dtype=torch.long), texts, lens, int(lens.max()), spk)
    out = host.CMTotalTTSSynthesize.from_model(model, T=T_steps, generator=Gen()).synthesize(batch)
    torch.cuda.synchronize()
    assert np.array_equal(_np(out[11]), g["mel_len"]) and np.array_equal(_np(out[10]), g["src_lens"])
    ok = _mask_near_pitch_flips(model, g, dict(texts=texts, src_lens=lens, spker_embeds=spk), "cmtts_VCTK")
    err = np.abs(_np(out[0]) - g[f"mel_T{T_steps}"])
    assert err[ok].max() < 1e-3, err[ok].max()


def test_synthesizer_reference_constructor_end_to_end(models, tmp_path):
    """CMTotalTTSSynthesize(model_path, model_step_num, args, preprocess_config, model_config, train_config) exactly as
    synthesize.py:35-86 builds it — checkpoint on disk, YAML dicts — must synthesise what the from_model form does."""
    import argparse
    from test_host_module_cpu import _reference_configs
    host = _host()
    g, cfg, sd, model = models("VCTK")
    pre, mod, tr = _reference_configs()
    (tmp_path / "CMDenoiserTTS").mkdir()
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "CMDenoiserTTS" / "model000300.pt")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        def __init__(self):
            self.i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[self.i]).to(DEV)
            self.i += 1
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    texts, lens, spk = torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]), torch.from_numpy(g["spker_embeds"])
    batch = (["a", "b", "c"], ["x", "y", "z"], torch.zeros(B, dtype=torch.long), texts, lens, int(lens.max()), spk)
    syn = host.CMTotalTTSSynthesize(str(tmp_path), 300, argparse.Namespace(T=4), pre, mod, tr, device=DEV, generator=Gen())
    out = syn.synthesize(batch)
    ref = host.CMTotalTTSSynthesize.from_model(model, T=4, generator=Gen()).synthesize(batch)
    host.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[11], ref[11]) and out[10] is batch[4]
    assert np.abs(_np(out[0]) - g["mel_T4"]).max() < 1e-3
    assert syn.model.to(DEV) is syn.model
    with pytest.raises(RuntimeError):
        syn.model.to("cpu")


def test_synth_samples_writes_the_reference_wav_files(models, tmp_path):
    """utils/tools.py:566-607: the synthesize flow ends in one 22 050 Hz int16 .wav per utterance under
    <path>/<restore_step>/ with the reference's file names; the samples are vocoder_infer's, trimmed to mel_len * hop."""
    import argparse
    from scipy.io import wavfile
    from test_host_module_cpu import _reference_configs
    host = _host()
    g, cfg, sd, model = models("VCTK")
    pre, mod, tr = _reference_configs()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=2))
    texts, lens, spk = torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]), torch.from_numpy(g["spker_embeds"])
    B = texts.shape[0]
    ids = [f"utt{i}" for i in range(B)]
    batch = (ids, ["x"] * B, torch.zeros(B, dtype=torch.long), texts, lens, int(lens.max()), spk)
    out = host.CMTotalTTSSynthesize.from_model(model, T=2).synthesize(batch)
    ref = host.vocoder_infer(out[0].transpose(1, 2), voc, mod, pre, lengths=(out[11] * cfg.hop_length).tolist())
    for mode, names in (("batch", [f"{i}.wav" for i in ids]), ("single", [f"{i}_p225.wav" for i in ids])):
        args = argparse.Namespace(restore_step=300, mode=mode, speaker_id="p225", teacher_forced=False)
        written = host.synth_samples(args, batch, out, voc, mod, pre, str(tmp_path), None)
        assert [os.path.relpath(w, tmp_path) for w in written] == [os.path.join("300", n) for n in names]
        for w, r, n in zip(written, ref, _np(out[11])):
            rate, data = wavfile.read(w)
            assert rate == 22050 and data.dtype == np.int16 and data.shape[0] == int(n) * cfg.hop_length
            assert np.array_equal(data, r)


def test_speaker_table_ids_are_checked_on_the_device():
    """ids into the speaker_emb table that already live on the GPU (the reference's to_device moved them) must
    raise like nn.Embedding does (model/cmtts.py:78), not be clamped to another speaker."""
    host = _host()
    cfg = get_config("VCTK_table")
    sd = synth_cmtts_state_dict(cfg, seed=1, dur_frames=3.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    texts = torch.randint(1, cfg.n_symbols, (2, 6))
    lens = torch.tensor([6, 6])
    model.duration_pitch_energy_net(torch.tensor([0, cfg.n_speaker - 1], device=DEV), texts, lens)
    for bad in ([0, cfg.n_speaker], [-1, 0]):
        with pytest.raises(IndexError):
            model.duration_pitch_energy_net(torch.tensor(bad, device=DEV), texts, lens)


def test_generic_denoise_path_matches_fused_sampler(models):
    """KarrasDenoiser.denoise around CMTotalTTS.forward (which re-runs the duration net on every call,
    tts_net.py:132-147) must give the onestep sample the fused cmtts_sample path gives."""
    host = _host()
    g, cfg, sd, model = models("LJSpeech")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 1)[0]
    x_T = torch.from_numpy(noise).to(DEV) * cfg.sigma_max
    sig = torch.full((B,), cfg.sigma_max, device=DEV)
    kw = dict(speakers=None, texts=torch.from_numpy(g["texts"]), src_lens=torch.from_numpy(g["src_lens"]), spker_embeds=None)
    _, den = host.KarrasDenoiser(distillation=True).denoise(model, x_T, sig, **kw)
    torch.cuda.synchronize()
    ok = _mask_near_pitch_flips(model, g, kw, "cmtts_LJSpeech")
    err = np.abs(_np(den[:, 0]) - g["mel_T1"])
    assert err[ok].max() < 1e-3, err[ok].max()


def test_host_side_samplers_match_fused(models):
    """§8(f) item 3: the reference's sampler loops run host-side over the same denoiser kernels; for the
    schedules synthesize.py uses they must reproduce the fused cmtts_sample result."""
    host = _host()
    g, cfg, sd, model = models("VCTK")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    cond = torch.from_numpy(g["cond"]).to(DEV)
    spk = torch.from_numpy(g["speaker_emb"]).to(DEV)
    diffusion = host.KarrasDenoiser(distillation=True)
    dist = host.make_distiller(diffusion, model, cond, spk)
    x_T = torch.from_numpy(noise[0]).to(DEV) * cfg.sigma_max
    sig = torch.tensor([cfg.sigma_max, 0.0], device=DEV)

    class Gen:
        i = 1

        def randn_like(self, x):
            t = torch.from_numpy(noise[Gen.i]).to(DEV)
            Gen.i += 1
            return t

    one = host.sample_onestep(dist, x_T, sig)[:, 0]
    assert np.abs(_np(one) - g["mel_T1"]).max() < 1e-3
    multi = host.stochastic_iterative_sampler(dist, x_T, sig, Gen(), ts=(0, 0, 1), steps=2)[:, 0]
    assert np.abs(_np(multi) - g["mel_T2"]).max() < 1e-3
    twice = host.our_multistep(dist, x_T, sig, T=2)
    assert torch.isfinite(twice).all() and tuple(twice.shape) == (B, 1, T, cfg.n_mels)


@pytest.mark.parametrize("sampler", ["euler", "heun", "dpm", "ancestral", "progdist"])
def test_ode_samplers_golden(models, golden, sampler):
    """§8(f) item 3: karras_sample_tts(sampler=euler|heun|dpm|ancestral|progdist) — the reference's other loops run
    host-side around the HIP denoiser — against the reference's own output for the same noise draws."""
    host = _host()
    g, cfg, sd, model = models("LJSpeech")
    gs = golden("samplers_LJSpeech")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        def __init__(self):
            self.i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[self.i]).to(DEV)
            self.i += 1
            assert tuple(t.shape) == tuple(shape)
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    gen = Gen()
    kwargs = dict(speakers=None, texts=torch.from_numpy(g["texts"]), src_lens=torch.from_numpy(g["src_lens"]))
    mel = host.karras_sample_tts(host.KarrasDenoiser(distillation=True), model, (B, 1, T, cfg.n_mels), steps=int(gs["steps_" + sampler]),
                                 model_kwargs=kwargs, sigma_min=cfg.sigma_min, sigma_max=cfg.sigma_max, rho=cfg.rho,
                                 sampler=sampler, generator=gen)
    torch.cuda.synchronize()
    assert gen.i == int(gs["draws_" + sampler])          # same number of noise draws as the reference
    ref = gs["mel_" + sampler]
    # fp32; the ODE steps divide by sigma, so the bound scales with the output (|mel| up to ~13 for heun)
    np.testing.assert_allclose(_np(mel), ref, atol=1e-3, rtol=2e-4)


def test_non_distilled_diffusion_routes(models, golden):
    """KarrasDenoiser(distillation=False): denoise uses get_scalings (karras_diffusion.py:81-85,395-398).  heun is
    checked against the reference run that way; onestep/multistep must then leave the fused cmtts_sample path
    (which bakes the boundary-condition scalings in) and differ from the distilled result."""
    host = _host()
    g, cfg, sd, model = models("LJSpeech")
    gs = golden("samplers_LJSpeech")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)

    class Gen:
        def __init__(self):
            self.i = 0

        def randn(self, *shape, **kw):
            t = torch.from_numpy(noise[self.i]).to(DEV)
            self.i += 1
            return t

        def randn_like(self, x):
            return self.randn(*x.shape)

    kwargs = dict(speakers=None, texts=torch.from_numpy(g["texts"]), src_lens=torch.from_numpy(g["src_lens"]))
    common = dict(model_kwargs=kwargs, sigma_min=cfg.sigma_min, sigma_max=cfg.sigma_max, rho=cfg.rho)
    edm = host.KarrasDenoiser(distillation=False)
    mel = host.karras_sample_tts(edm, model, (B, 1, T, cfg.n_mels), steps=int(gs["steps_heun"]), sampler="heun",
                                 generator=Gen(), **common)
    np.testing.assert_allclose(_np(mel), gs["mel_heun_edm"], atol=3e-3, rtol=2e-4)
    # onestep at sigma_max: c_skip/c_out of the two scalings differ by O(sigma_min/sigma_max) only, but they differ
    one_edm = host.karras_sample_tts(edm, model, (B, 1, T, cfg.n_mels), steps=2, sampler="onestep", generator=Gen(), **common)
    one_cm = host.karras_sample_tts(host.KarrasDenoiser(distillation=True), model, (B, 1, T, cfg.n_mels), steps=2, sampler="onestep",
                                    generator=Gen(), **common)
    torch.cuda.synchronize()
    assert np.abs(_np(one_cm) - g["mel_T1"]).max() < 1e-3
    d = np.abs(_np(one_edm) - _np(one_cm)).max()
    assert 0 < d < 1e-2, d
    # a multistep schedule the fused path does not cover runs through the host loop instead of raising
    gen = Gen()
    ms = host.karras_sample_tts(host.KarrasDenoiser(distillation=True), model, (B, 1, T, cfg.n_mels), steps=4, sampler="multistep",
                                ts=(0, 1, 3), generator=gen, **common)
    torch.cuda.synchronize()
    assert gen.i == int(gs["draws_multistep_ts013"])
    assert np.abs(_np(ms) - gs["mel_multistep_ts013"]).max() < 1e-3


def _check_variance_gpu(out, gc, tag):
    np.testing.assert_allclose(_np(out["log_d_predictions"]), gc[tag + "_log_d"], atol=5e-5)
    np.testing.assert_array_equal(_np(out["d_rounded"]), gc[tag + "_d_rounded"])            # bit-exact
    np.testing.assert_array_equal(_np(out["mel_lens"]), gc[tag + "_mel_len"])
    np.testing.assert_allclose(_np(out["e_predictions"]), gc[tag + "_e_pred"], atol=1e-4)
    pp = out["p_predictions"]
    np.testing.assert_allclose(_np(pp["cwt"]), gc[tag + "_cwt_out"], atol=3e-4)
    np.testing.assert_allclose(_np(pp["f0_denorm"]), gc[tag + "_f0_denorm"], rtol=3e-4, atol=2e-3)
    same = pitch_flips(_np(pp["p_idx"]), gc[tag + "_p_idx"], gc[tag + "_f0_denorm"], "controls_VCTK:" + tag)
    err = np.abs(_np(out["cond"]) - gc[tag + "_cond"])[same].max()
    assert err < 1e-3, err


def test_variance_controls_golden(models, golden):
    """p_control / e_control / d_control (model/modules.py:270,326,369) against the reference's output."""
    g, cfg, sd, model = models("VCTK")
    gc = golden("controls_VCTK")
    out = model.duration_pitch_energy_net(None, torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]),
                                          spker_embeds=torch.from_numpy(g["spker_embeds"]),
                                          p_control=float(gc["p_control"]), e_control=float(gc["e_control"]),
                                          d_control=float(gc["d_control"]))
    torch.cuda.synchronize()
    _check_variance_gpu(out, gc, "ctl")
    # the settings do not stick: the next plain call reproduces the inference golden
    plain = model.duration_pitch_energy_net(None, torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]),
                                            spker_embeds=torch.from_numpy(g["spker_embeds"]))
    np.testing.assert_array_equal(_np(plain["d_rounded"]), g["d_rounded"])
    np.testing.assert_array_equal(_np(plain["e_idx"]), g["e_idx"])


def test_variance_teacher_forced_golden(models, golden):
    """Teacher-forced duration / energy / pitch targets (model/modules.py:318-328,365-367,379-390)."""
    g, cfg, sd, model = models("VCTK")
    gc = golden("controls_VCTK")
    T = gc["tf_cwt_spec"].shape[1]
    pt = {k: torch.from_numpy(gc["tf_" + k]) for k in ("cwt_spec", "f0_mean", "f0_std", "uv")}
    out = model.duration_pitch_energy_net(None, torch.from_numpy(g["texts"]), torch.from_numpy(g["src_lens"]),
                                          mels=torch.zeros(len(g["src_lens"]), 1, T, cfg.n_mels),
                                          p_targets=pt, e_targets=torch.from_numpy(gc["tf_e_target"]),
                                          d_targets=torch.from_numpy(gc["tf_d_target"]),
                                          spker_embeds=torch.from_numpy(g["spker_embeds"]))
    torch.cuda.synchronize()
    _check_variance_gpu(out, gc, "tf")


def test_hifigan_golden(golden):
    host = _host()
    g = golden("hifigan")
    hcfg = HifiGanConfig()
    hsd = synth_hifigan_state_dict(hcfg, seed=int(g["seed"]))
    voc = host.Generator(hcfg, DEV).load_state_dict(hsd)
    mel_ct = torch.from_numpy(np.ascontiguousarray(g["mel"].transpose(0, 2, 1)))
    wav = voc(mel_ct)
    torch.cuda.synchronize()
    assert tuple(wav.shape) == tuple(g["wav"].shape)
    err = np.abs(_np(wav) - g["wav"]).max()
    assert err < 1e-4, err
    pcm = host.vocoder_infer(mel_ct, voc, lengths=g["mel_lens"] * 256)
    for i, name in enumerate(["pcm0", "pcm1"]):
        assert pcm[i].dtype == np.int16 and pcm[i].shape == g[name].shape
        assert np.abs(pcm[i].astype(np.int32) - g[name].astype(np.int32)).max() <= 4


@pytest.mark.parametrize("B,T", [(1, 1), (2, 7), (3, 65), (5, 129), (1, 700), (33, 513)])
def test_vocoder_winograd_odd_shapes(B, T):
    """VERDICT r04 #5a for the generator: the Winograd forms of the wide-stage convs work on output PAIRS one dilation apart (conv_xlw_kernel:
    30- / 32-pair tiles) or, for dilation 1 since round 5, on QUADS of outputs (conv_xlq_kernel: 16-quad tiles, F(4,3) tap groups) — a
    one-frame mel, lengths that leave a lone column / pair / quad in the last tile at every dilation, an odd batch.  Both forced onto every
    shape (voc_wino = 2; voc_wino43 = 1 | 0), against the direct form: fp32 rounding only."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=22))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(7 * T + B)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_wino", 2)
    prev43 = _lib.internal_set(b"voc_wino43", 3)      # F(4,3) at every dilation and width (the default takes dilation 3 / 5 in that form only at C = 256)
    try:
        got = voc(mel).clone()
        _lib.internal_set(b"voc_wino43", 1)
        got_default = voc(mel).clone()
        prevq = _lib.internal_set(b"voc_qpair", 0)         # round 6: without the fused k = 3 pairs (C = 128: two conv_xlq launches; C = 64: the direct pair kernel)
        got_noq = voc(mel).clone()
        _lib.internal_set(b"voc_qpair", prevq)
        _lib.internal_set(b"voc_wino43", 0)
        got23 = voc(mel).clone()
        _lib.internal_set(b"voc_wino", 0)
        ref = voc(mel).clone()
    finally:
        _lib.internal_set(b"voc_wino", prev)
        _lib.internal_set(b"voc_wino43", prev43)
    torch.cuda.synchronize()
    d, d23, dd = float((got - ref).abs().max()), float((got23 - ref).abs().max()), float((got_default - ref).abs().max())
    report(f"VOC_WINOGRAD_ODD B={B} T={T}: max|d wav| vs the direct form: F(4,3) everywhere {d:.2e}, the default mix {dd:.2e}, F(2,3) tap groups everywhere {d23:.2e}")
    assert got.shape == ref.shape == (B, 1, T * 256) and torch.isfinite(got).all() and torch.isfinite(got23).all() and torch.isfinite(got_default).all()
    assert 0 < d <= VOC_WINO_TOL and 0 < d23 <= VOC_WINO_TOL and 0 < dd <= VOC_WINO_TOL, (d, d23, dd)
    assert not torch.equal(got, got23)
    dq = float((got_default - got_noq).abs().max())
    assert 0 < dq <= VOC_WINO_TOL, dq          # (the k = 3 pairs of the C = 64 / 128 stages change form: fp32 Winograd rounding)


@pytest.mark.parametrize("B,T", [(24, 350), (9, 1000), (32, 512)])
def test_vocoder_winograd_vs_direct(B, T):
    """conv_xlw_kernel (round 4): the ResBlock convs of the C = 256 / 128 stages as Winograd convolutions over output pairs (t, t + dilation):
    k = 3 / 7 / 11 as F(2,3) groups + an F(2,2) / single-tap remainder, all three dilations, residual and MRF accumulation, ragged last tiles
    (T not a multiple of the 64- / 60-column tiles); conv_xlq_kernel (round 5, the default for dilation 1): F(4,3) groups over output quads.
    Against the direct form on a chip-filling batch: fp32 rounding only; and the reference's golden wav with the form forced onto its
    small shape."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=21))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(T + B)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_wino", 1)
    try:
        got = voc(mel).clone()
        _lib.internal_set(b"voc_wino", 0)
        ref = voc(mel).clone()
        _lib.internal_set(b"voc_wino", 2)
        one = voc(mel[:1]).clone()             # forced onto one utterance: every conv of both wide stages in the Winograd form
    finally:
        _lib.internal_set(b"voc_wino", prev)
    torch.cuda.synchronize()
    d = float((got - ref).abs().max())
    report(f"VOC_WINOGRAD B={B} T={T}: max|d wav| vs the direct form {d:.2e}; one utterance forced {float((one[0] - ref[0]).abs().max()):.2e}")
    assert torch.isfinite(got).all() and 0 < d <= VOC_WINO_TOL
    assert float((one[0] - ref[0]).abs().max()) <= VOC_WINO_TOL


def test_hifigan_golden_winograd_forced(golden):
    """The reference's own wav for the golden mel with the Winograd form forced onto its small shape (the default takes it from 1024 column
    tiles on): the same 1e-4 bound as test_hifigan_golden, PCM within 4 LSB."""
    host = _host()
    g = golden("hifigan")
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=int(g["seed"])))
    mel_ct = torch.from_numpy(np.ascontiguousarray(g["mel"].transpose(0, 2, 1)))
    prev = _lib.internal_set(b"voc_wino", 2)
    try:
        wav = voc(mel_ct)
        pcm = host.vocoder_infer(mel_ct, voc, lengths=g["mel_lens"] * 256)
    finally:
        _lib.internal_set(b"voc_wino", prev)
    torch.cuda.synchronize()
    assert np.abs(_np(wav) - g["wav"]).max() < 1e-4
    for i, name in enumerate(["pcm0", "pcm1"]):
        assert np.abs(pcm[i].astype(np.int32) - g[name].astype(np.int32)).max() <= 4


def test_fastspeech_decoder_golden_and_long(golden):
    """FastspeechDecoder over the frame axis on the encoder's FFT-block kernels: the reference module's output for
    the golden input (both mask forms), then T = 700 ragged frames against the oracle (attention over 700 keys)."""
    from cmtts_amd.weights import synth_decoder_state_dict
    host = _host()
    g = golden("decoder_LJSpeech")
    cfg = get_config("LJSpeech")
    sd = synth_cmtts_state_dict(cfg, seed=2)
    dsd = synth_decoder_state_dict(cfg, seed=int(g["seed"]))
    sd.update(dsd)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    x, lens = g["x"], g["lens"]
    pad = np.arange(x.shape[1])[None, :] >= lens[:, None]
    xz = x.copy()
    xz[pad] = 0
    y = model.decoder(torch.from_numpy(xz))
    np.testing.assert_allclose(_np(y), g["y_auto"], atol=5e-5)
    y = model.decoder(torch.from_numpy(x), torch.from_numpy(pad))
    np.testing.assert_allclose(_np(y), g["y_mask"], atol=5e-5)
    rs = np.random.RandomState(1)
    B, T = 2, 700
    lens = np.asarray([700, 431])
    xl = rs.standard_normal(size=(B, T, cfg.hidden)).astype(np.float32)
    padl = np.arange(T)[None, :] >= lens[:, None]
    xl[padl] = 0
    xl[0, 5, 0] = 0.0          # a valid frame whose channel 0 is exactly zero: position 0 there, later positions shift
    ref = O.fastspeech_decoder(dsd, cfg, xl, padl)
    y = model.decoder(torch.from_numpy(xl), torch.from_numpy(padl))
    np.testing.assert_allclose(_np(y), ref, atol=1e-4)
    prev = _lib.internal_set(b"attn_fused", 0)          # T = 700 takes the fused (key-chunked) attention kernel since round 3
    try:
        y3 = model.decoder(torch.from_numpy(xl), torch.from_numpy(padl))
    finally:
        _lib.internal_set(b"attn_fused", prev)
    d = float((y3 - y).abs().max())
    assert 0 < d < 5e-5, d
    # a model without decoder.* tensors refuses loudly
    plain = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=2))
    with pytest.raises(RuntimeError):
        plain.decoder(torch.from_numpy(xz))
    with pytest.raises(NotImplementedError):
        holes = pad.copy(); holes[0, 3] = True
        model.decoder(torch.from_numpy(x), torch.from_numpy(holes))


@pytest.mark.parametrize("variant,B,L", [("VCTK", 3, 40), ("LJSpeech", 32, 85)])
def test_branch_streams_bitwise(variant, B, L):
    """Independent branches (energy predictor, V projection, cwt statistics MLP, conditioner GEMM) run on a side stream
    forked from / joined into the caller's stream: the same kernels, so every output must be bit-identical to the
    in-line order, call after call (a missing dependency would show up as a race)."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=9, dur_frames=5.0, dur_spread=0.02)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(B)
    lens = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    T = 6 * L
    noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run():
        out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=T)
        mel = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], 2, noise)
        torch.cuda.synchronize()
        return [out[k].clone() for k in ("cond_ct", "log_d_predictions", "e_predictions", "mel_lens", "mel2ph")] + \
               [out["p_predictions"]["cwt"].clone(), out["p_predictions"]["f0_mean"].clone(), mel.clone()]

    prev = lib.cmtts_set_option(b"branch_streams", 0)
    try:
        ref = run()
        lib.cmtts_set_option(b"branch_streams", 1)
        for _ in range(4):
            got = run()
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
    finally:
        lib.cmtts_set_option(b"branch_streams", prev)


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 3, 171), ("VCTK", 4, 33), ("LJSpeech", 2, 1), ("LJSpeech", 2, 192),
                                         ("LJSpeech", 3, 193), ("VCTK", 2, 640), ("LJSpeech", 1, 1000)])
def test_fused_attention_matches_three_launch_path(variant, B, L):
    """attention.hip (QKV projection as one contraction + softmax(q k^T / sqrt(dh) + key mask) v in one launch, scores in
    registers) against the three-launch path (K^T Q GEMM -> softmax_cols -> V P^T GEMM): the same operations in a
    different fp32 summation order -> encoder output within 2e-5, integer stages identical; ragged lengths exercise the
    key mask, L = 171 / 192 the six-wave form, L = 1 the degenerate one; L > 192 (up to max_seq_len = 1000,
    config/LJSpeech/model.yaml:55) the key-chunked online-softmax kernel with the queries split over workgroups."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=19, dur_frames=3.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(L)
    lens = np.maximum((rs.uniform(0.3, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    prev = _lib.internal_set(b"attn_fused", 0)
    try:
        ref = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        _lib.internal_set(b"attn_fused", 1)
        got = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        again = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"attn_fused", prev)
    assert torch.isfinite(got["enc_out"]).all()
    err = float((got["enc_out"] - ref["enc_out"]).abs().max())
    assert err < 2e-5, err
    assert err > 0 or L == 1                     # it really took the other path
    assert torch.equal(got["mel_lens"], ref["mel_lens"]) and torch.equal(got["mel2ph"], ref["mel2ph"])
    assert torch.equal(again["enc_out"], got["enc_out"])        # deterministic
    if L == 33:   # one utterance alone: bit-identical to its rows in the batch
        one = model.duration_pitch_energy_net(None, torch.from_numpy(texts[1:2, :int(lens[1])]), torch.from_numpy(lens[1:2]),
                                              spker_embeds=None if spk is None else spk[1:2], max_mel_len=3 * L)
        n = int(lens[1])
        if n == L:
            assert torch.equal(one["enc_out"][0, :n], got["enc_out"][1, :n])


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 5, 128), ("LJSpeech", 3, 97), ("VCTK", 7, 33), ("LJSpeech", 2, 32), ("VCTK", 1, 25),
                                         ("LJSpeech", 4, 1), ("LibriTTS", 9, 64)])
def test_attention_qb_bitwise(variant, B, L):
    """Round 6: attention_qb_kernel (attention.hip: a workgroup = one (utterance, head, block of 32 queries); key tiles over the waves, the
    un-normalised probabilities through LDS, summed in attention_kernel's order; output channels over the waves) against attention_kernel
    (a workgroup = one (utterance, head)): every accumulation chain has the same operands in the same order => the whole text side bit for
    bit, at one to four key tiles, ragged lengths (key mask, a fully padded query block), L = 1."""
    host = _host()
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=23, dur_frames=3.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(1000 + L)
    lens = np.maximum((rs.uniform(0.2, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    prev = _lib.internal_set(b"attn_qb", 0)
    try:
        ref = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        ref = {k: ref[k].clone() for k in ("enc_out", "mel_lens", "mel2ph", "cond_ct")}
        assert _lib.internal_set(b"attn_qb", 1) == 0
        got = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"attn_qb", prev)
    assert torch.isfinite(got["enc_out"]).all()
    for k in ref:
        assert torch.equal(got[k], ref[k]), (k, float((got[k].float() - ref[k].float()).abs().max()))


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 3, 40), ("LibriTTS", 1, 7)])
def test_stats_mlp_and_transposed_factor_bitwise(variant, B, L):
    """Round 6, two launch-count changes on the frame-level half that must not move a bit: (a) cwt_stats_layers (model/modules.py:212-215) as ONE
    launch (kernels.hip: stats_mlp_kernel — dense_small_kernel<4>'s K slices and sum order) against the three dense_small launches: f0 statistics,
    pitch buckets and conditioning equal; (b) the conditioner factor's channel-contiguous copy produced by cmtts_frame_forward_sub_t on the branch
    stream (CondFactors.p1t) against the sampler transposing cond_p1 at its entry (cmtts_sample_factored): the same mel."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=11, dur_frames=6.0, dur_spread=0.0))
    rs = np.random.RandomState(77 + L)
    texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64))
    lens = torch.full((B,), L, dtype=torch.int64)
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    T = 6 * L
    prev = _lib.internal_set(b"stats_mlp", 0)
    try:
        ref = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=T)
        ref = {"cond_ct": ref["cond_ct"].clone(), "p_idx": ref["p_predictions"]["p_idx"].clone(), "f0_mean": ref["p_predictions"]["f0_mean"].clone(),
               "f0_std": ref["p_predictions"]["f0_std"].clone()}
        assert _lib.internal_set(b"stats_mlp", 1) == 0
        out = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=T)
    finally:
        _lib.internal_set(b"stats_mlp", prev)
    assert torch.equal(out["p_predictions"]["f0_mean"], ref["f0_mean"]) and torch.equal(out["p_predictions"]["f0_std"], ref["f0_std"])
    assert torch.equal(out["p_predictions"]["p_idx"], ref["p_idx"]) and torch.equal(out["cond_ct"], ref["cond_ct"])
    fac = out["cond_factors"]
    assert fac is not None and fac.p1t is not None
    assert torch.equal(fac.p1t, fac.p1.view(B, cfg.res_layers, cfg.res_channels, fac.p1_ld).transpose(2, 3).contiguous())
    noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(4)).to(DEV)
    spk_emb = out.get("speaker_emb")
    prev_p = _lib.load().cmtts_set_persistent_denoiser(2)
    try:
        mel_t = host.sample_with_cond(model, out["cond_ct"], spk_emb, 2, noise).clone()
        plain = host.CondFactors(fac.p1, fac.p1_ld, fac.L, fac.mel2ph, fac.p_idx, out["cond_ct"])      # no p1t: the sampler transposes
        mel_s = host.sample_with_cond(model, out["cond_ct"], spk_emb, 2, noise, factors=plain).clone()
    finally:
        _lib.load().cmtts_set_persistent_denoiser(prev_p)
    host.synchronize()
    assert torch.isfinite(mel_t).all() and torch.equal(mel_t, mel_s)


@pytest.mark.parametrize("variant,B,L,mode", [("LJSpeech", 32, 85, "plain"), ("VCTK", 3, 40, "control"), ("LibriTTS", 2, 171, "target"), ("LJSpeech", 1, 7, "plain")])
def test_energy_head_bitwise(variant, B, L, mode):
    """Round 6: get_energy_embedding + the embedding add (model/modules.py:318-328,358-363) inside the energy predictor's head launch
    (kernels.hip: ln_linear_kernel<1, true>) against ln_linear + energy_embed_kernel behind the join: the same bucketize and the same add per
    element — prediction, buckets, conditioning and everything downstream bit for bit; with an energy control (the returned prediction is the
    scaled one) and with teacher-forced energy targets; ragged lengths."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=29, dur_frames=5.0, dur_spread=0.0))
    rs = np.random.RandomState(300 + L)
    lens_np = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1)
    lens_np[0] = L
    texts_np = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts_np[np.arange(L)[None, :] >= lens_np[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    kw = {}
    if mode == "control":
        kw["e_control"] = 1.3
    if mode == "target":
        kw["e_targets"] = torch.from_numpy(rs.uniform(-1.0, 4.0, size=(B, L)).astype(np.float32))
    run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts_np), torch.from_numpy(lens_np), spker_embeds=spk, max_mel_len=5 * L, **kw)
    keys = ("e_predictions", "e_idx", "cond_ct", "mel2ph")
    prev = _lib.internal_set(b"energy_head", 0)
    try:
        ref = run()
        ref = {k: ref[k].clone() for k in keys} | {"p_idx": ref["p_predictions"]["p_idx"].clone()}
        assert _lib.internal_set(b"energy_head", 1) == 0
        got = run()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"energy_head", prev)
    for k in keys:
        assert torch.equal(got[k], ref[k]), (k, mode)
    assert torch.equal(got["p_predictions"]["p_idx"], ref["p_idx"])


@pytest.mark.parametrize("B,T", [(3, 96), (2, 77), (32, 512), (1, 5)])
def test_fused_input_projection_bitwise(B, T):
    """inproj.hip (c_in scaling + [B,T,80] -> [B,80,T] + relu(input_projection) + clearing of the persistent kernel's halo
    granules in one launch) against mel_prep + the generic conv kernel + hipMemsetAsync: same products in the same order and
    the same epilogue -> the sampler output must not change by a bit (T = 77 / 5: ragged and sub-tile utterances; B = 32 x 512
    takes the persistent stack, whose halo buffer the fused kernel clears)."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=33))
    gen = torch.Generator(device="cpu").manual_seed(T)
    cond = torch.randn(B, cfg.hidden, T, generator=gen).to(DEV)
    noise = torch.randn(5, B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    prev = _lib.internal_set(b"inproj_fused", 0)
    try:
        ref = host.sample_with_cond(model, cond, None, 4, noise).clone()
        _lib.internal_set(b"inproj_fused", 1)
        got = host.sample_with_cond(model, cond, None, 4, noise).clone()
        again = host.sample_with_cond(model, cond, None, 4, noise).clone()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"inproj_fused", prev)
    assert torch.isfinite(ref).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert torch.equal(again, ref)
    host.check_async_error()


@pytest.mark.parametrize("variant", ["LJSpeech", "VCTK"])
def test_step_embedding_cache_bitwise(variant):
    """cmtts_sample keeps the timestep-only part of the step embedding (DiffusionEmbedding -> MLP -> 20 diffusion projections)
    per rescaled timestep on the device and re-uses it in later calls: a cache hit must give the bits of the computation it
    replaces — single-speaker and with the per-utterance speaker projection added on top — and so must the first (filling) call."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=31))
    gen = torch.Generator(device="cpu").manual_seed(3)
    B, T = 3, 96
    cond = torch.randn(B, cfg.hidden, T, generator=gen).to(DEV)
    spk = torch.randn(B, cfg.hidden, generator=gen).to(DEV) if cfg.multi_speaker else None
    outs = {}
    prev = lib.cmtts_set_option(b"step_cache", 0)
    try:
        for n_steps in (1, 4):
            noise = torch.randn(n_steps + 1 if n_steps > 1 else 1, B, 1, T, cfg.n_mels, generator=gen).to(DEV)
            lib.cmtts_set_option(b"step_cache", 0)
            ref = host.sample_with_cond(model, cond, spk, n_steps, noise).clone()
            lib.cmtts_set_option(b"step_cache", 1)
            first = host.sample_with_cond(model, cond, spk, n_steps, noise).clone()      # fills the cache (or hits it: T = 4 after T = 1)
            torch.cuda.synchronize()
            hit = host.sample_with_cond(model, cond, spk, n_steps, noise).clone()        # the copy has completed: a hit
            torch.cuda.synchronize()
            outs[n_steps] = (ref, first, hit)
    finally:
        lib.cmtts_set_option(b"step_cache", prev)
    for n_steps, (ref, first, hit) in outs.items():
        assert torch.isfinite(ref).all()
        assert torch.equal(first, ref), (n_steps, float((first - ref).abs().max()))
        assert torch.equal(hit, ref), (n_steps, float((hit - ref).abs().max()))


def test_predictor_conv_xl_bitwise(models):
    """The frame-level 256 -> 256 predictor conv (cwt predictor, k = 5, ReLU) on conv_xl_kernel (x tile + halo resident in
    LDS, weights streamed as MFMA A fragments) keeps the generic kernel's accumulation order and epilogue: not a bit changes
    (B = 32 x 512 frames takes it, the small golden batches the generic kernel)."""
    host = _host()
    lib = _lib.load()
    g, cfg, sd, model = models("LJSpeech")
    rs = np.random.RandomState(5)
    B, L = 32, 85
    lens = rs.randint(60, L + 1, size=B).astype(np.int64)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), max_mel_len=512)
    prev = _lib.internal_set(b"pred_xl", 1)
    try:
        one = run()
        _lib.internal_set(b"pred_xl", 0)
        ref = run()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"pred_xl", prev)
    assert torch.equal(one["p_predictions"]["cwt"], ref["p_predictions"]["cwt"])
    assert torch.equal(one["cond"], ref["cond"]) and torch.equal(one["p_predictions"]["p_idx"], ref["p_predictions"]["p_idx"])


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 3, 40), ("LibriTTS", 1, 7)])
def test_predictor_head_fused_matches_two_launch_path(variant, B, L):
    """ln_linear_kernel (last LayerNorm of a predictor + its linear head in one launch, reductions by wave shuffles) against
    layernorm_ct + chan_linear: the same operations in another fp32 summation order -> predictions within 1e-5, integer
    stages (durations, energy / pitch buckets, mel2ph) identical; masked phonemes give exactly 0 in the duration head; the
    wave-parallel durations kernel gives the serial one's cumulative sums (checked through mel_lens / mel2ph)."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=23, dur_frames=3.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(100 + L)
    lens = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
    prev = _lib.internal_set(b"pred_head", 0)
    try:
        ref = run()
        _lib.internal_set(b"pred_head", 1)
        got = run()
        again = run()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"pred_head", prev)
    for k in ("log_d_predictions", "e_predictions"):
        err = float((got[k] - ref[k]).abs().max())
        assert err < 1e-5, (k, err)
        assert torch.equal(again[k], got[k])
    err = float((got["p_predictions"]["cwt"] - ref["p_predictions"]["cwt"]).abs().max())
    assert 0 < err < 1e-5, err
    pad = torch.arange(L)[None, :] >= torch.from_numpy(lens)[:, None]
    assert float(got["log_d_predictions"].cpu()[pad].abs().max() if pad.any() else 0.0) == 0.0
    for k in ("d_rounded", "mel_lens", "mel2ph", "e_idx"):
        assert torch.equal(got[k], ref[k]), k
    flips = int((got["p_predictions"]["p_idx"] != ref["p_predictions"]["p_idx"]).sum())
    assert flips == 0, flips
    assert float((got["cond"] - ref["cond"]).abs().max()) < 1e-5


def test_length_mask_kernel():
    """get_mask_from_lengths utils/tools.py:275-283 (True = padding): bit-exact against arange >= len."""
    host = _host()
    for lens, W in (([3, 0, 7], 7), ([1], 1), ([1026, 540, 798, 240], 1024), (list(range(1, 258)), 300)):
        l = torch.tensor(lens, dtype=torch.int64, device=DEV)
        m = host.get_mask_from_lengths(l, W)
        ref = torch.arange(W, device=DEV)[None, :] >= l[:, None]
        assert m.dtype == torch.bool and torch.equal(m, ref)
    assert torch.equal(host.get_mask_from_lengths(torch.tensor([2, 5], device=DEV)),
                       torch.tensor([[0, 0, 1, 1, 1], [0, 0, 0, 0, 0]], dtype=torch.bool, device=DEV))


def test_wav_to_int16_wraps_like_numpy():
    lib = _lib.load()
    w = np.float32([1.0, -1.0, 0.99999, -0.00002, 0.5, -0.5])
    wd = torch.from_numpy(w).to(DEV)
    pcm = torch.empty(w.shape, dtype=torch.int16, device=DEV)
    _lib.check(lib.cmtts_wav_to_int16(wd.data_ptr(), pcm.data_ptr(), w.size, 32768.0, None))
    torch.cuda.synchronize()
    assert _np(pcm).tolist() == O.wav_to_int16(w).tolist() == [-32768, -32768, 32767, 0, 16384, -16384]


# ----------------------------------------------------------------------------- vs oracle, larger/other shapes

def test_full_path_vs_oracle_random_batch():
    """B=4 ragged phoneme lengths, fresh seed, T = 2: HIP path vs numpy oracle end to end
    (integer stages bit-exact wherever the oracle's own decision margin is comfortable)."""
    host = _host()
    cfg = get_config("VCTK")
    sd = synth_cmtts_state_dict(cfg, seed=11, dur_frames=5.0, dur_spread=0.02)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(5)
    B, L = 4, 33
    lens = np.asarray([33, 21, 30, 5], np.int64)
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)
    st = O.duration_pitch_speaker_net(sd, cfg, texts, lens, spk)
    out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens),
                                          spker_embeds=torch.from_numpy(spk))
    torch.cuda.synchronize()
    np.testing.assert_allclose(_np(out["log_d_predictions"]), st["log_d"], atol=5e-5)
    pre = np.exp(st["log_d"].astype(np.float64)) - 1
    valid = np.arange(L)[None, :] < lens[:, None]
    safe = (np.abs(pre - np.floor(pre) - 0.5) > 1e-3) | ~valid
    assert np.array_equal(_np(out["d_rounded"])[safe], st["d_rounded"][safe])
    if safe.all():
        assert np.array_equal(_np(out["mel_lens"]), st["mel_len"])
        assert np.array_equal(_np(out["mel2ph"]), st["mel2ph"])
    T = st["cond"].shape[1]
    noise = np.stack([rs.standard_normal(size=(B, 1, T, cfg.n_mels)).astype(np.float32) for _ in range(3)])
    cond_ct = torch.from_numpy(np.ascontiguousarray(st["cond"].transpose(0, 2, 1))).to(DEV)
    mel = host.sample_with_cond(model, cond_ct, torch.from_numpy(st["speaker_emb"]).to(DEV), 2,
                                torch.from_numpy(noise).to(DEV))
    torch.cuda.synchronize()
    ref = O.karras_sample_tts(sd, cfg, st["cond"], st["speaker_emb"], 2, list(noise))
    assert np.abs(_np(mel) - ref).max() < 1e-3


def test_smallest_inputs_vs_oracle():
    """One utterance of ONE phoneme (a handful of frames: every conv is all halo, every tile is ragged, attention is
    1x1) and a 2-utterance batch of lengths (1, 2), end to end against the oracle, T = 1 and 4."""
    host = _host()
    cfg = get_config("LJSpeech")
    sd = synth_cmtts_state_dict(cfg, seed=4, dur_frames=3.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(8)
    for lens in ([1], [1, 2]):
        lens = np.asarray(lens, np.int64)
        B, L = len(lens), int(lens.max())
        texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
        texts[np.arange(L)[None, :] >= lens[:, None]] = 0
        st = O.duration_pitch_speaker_net(sd, cfg, texts, lens, None)
        out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens))
        torch.cuda.synchronize()
        assert np.array_equal(_np(out["d_rounded"]), st["d_rounded"])
        assert np.array_equal(_np(out["mel_lens"]), st["mel_len"]) and st["mel_len"].min() >= 1
        assert np.array_equal(_np(out["mel2ph"]), st["mel2ph"])
        T = st["cond"].shape[1]
        assert T <= 8
        noise = np.stack([rs.standard_normal(size=(B, 1, T, cfg.n_mels)).astype(np.float32) for _ in range(5)])
        for n_steps in (1, 4):
            mel = host.sample_with_cond(model, out["cond_ct"], None, n_steps, torch.from_numpy(noise).to(DEV))
            torch.cuda.synchronize()
            ref = O.karras_sample_tts(sd, cfg, _np(out["cond"]), None, n_steps, list(noise))
            assert np.abs(_np(mel) - ref).max() < 1e-3
        np.testing.assert_allclose(_np(out["cond"]), st["cond"], atol=1e-4)


def test_bucketed_ragged_shard_vs_oracle():
    """BASELINE configs[3] shape of ONE rank: LibriTTS model (multi-speaker, no uv), ragged phoneme lengths,
    frames padded to the static 1024 bucket (the longest utterance is truncated by the bucket, as the
    reference's pad() does with mels=x), T = 4.  Integer stages bit-exact, mel < 1e-3 vs the oracle."""
    host = _host()
    cfg = get_config("LibriTTS")
    sd = synth_cmtts_state_dict(cfg, seed=21, dur_frames=6.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(3)
    lens = np.asarray([171, 90, 133, 40], np.int64)
    B, L, T = len(lens), 171, 1024
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)
    st = O.duration_pitch_speaker_net(sd, cfg, texts, lens, spk, max_mel_len=T)
    out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens),
                                          spker_embeds=torch.from_numpy(spk), max_mel_len=T)
    torch.cuda.synchronize()
    assert st["mel_len"].tolist() == [1026, 540, 798, 240]
    assert np.array_equal(_np(out["mel_lens"]), st["mel_len"])
    assert np.array_equal(_np(out["d_rounded"]), st["d_rounded"])
    ref_m2p = O.dur_to_mel2ph(st["d_rounded"], st["src_mask"], T)
    assert np.array_equal(_np(out["mel2ph"]), ref_m2p)
    valid = np.arange(L)[None, :] < lens[:, None]
    e_same = (_np(out["e_idx"]) == st["e_idx"]) | ~valid
    p_same = _np(out["p_predictions"]["p_idx"]) == st["p_idx"]
    # vs the numpy oracle on unsearched inputs (no margin search as for the goldens): the bucket decisions that differ
    # are counted and pinned (measured on MI355X, round 2), not averaged away
    n_e, n_p = int((~e_same).sum()), int((~p_same).sum())
    report(f"BUCKET_FLIPS bucketed LibriTTS shard vs oracle: energy {n_e} of {int(valid.sum())} phonemes, pitch {n_p} of {p_same.size} frames")
    assert n_e <= KNOWN_ORACLE_FLIPS["energy"] and n_p <= KNOWN_ORACLE_FLIPS["pitch"]
    from conftest import FLIP_MARGIN
    on_boundary = ~pitch_margin_mask(st["f0_denorm"], FLIP_MARGIN)
    assert not (~p_same & ~on_boundary).any(), "a pitch bucket differs away from a rounding boundary"
    assert (np.abs(_np(out["p_predictions"]["p_idx"]) - st["p_idx"]) <= 1).all()
    # frames fed by agreeing energy buckets and pitch buckets must match the oracle's conditioning
    ph = np.clip(ref_m2p - 1, 0, L - 1)
    ok = p_same & np.take_along_axis(e_same, ph, 1)
    np.testing.assert_allclose(_np(out["cond"])[ok], st["cond"][ok], atol=1e-4)
    noise = np.stack([rs.standard_normal(size=(B, 1, T, cfg.n_mels)).astype(np.float32) for _ in range(5)])
    cond_ct = torch.from_numpy(np.ascontiguousarray(st["cond"].transpose(0, 2, 1))).to(DEV)
    mel = host.sample_with_cond(model, cond_ct, torch.from_numpy(st["speaker_emb"]).to(DEV), 4,
                                torch.from_numpy(noise).to(DEV))
    torch.cuda.synchronize()
    ref = O.karras_sample_tts_torch(sd, cfg, st["cond"], st["speaker_emb"], 4, list(noise))
    assert np.abs(_np(mel) - ref).max() < 1e-3


def test_end_to_end_wav_multispeaker_batch(voc_form):
    """BASELINE configs[2]/[4] shape in fp32: VCTK model, T = 2, universal-vocoder architecture, int16 out.
    Every utterance of the batch must be bit-identical to synthesising it in a batch of its own padding."""
    host = _host()
    cfg = get_config("VCTK")
    sd = synth_cmtts_state_dict(cfg, seed=8, dur_frames=5.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=8))
    rs = np.random.RandomState(4)
    B, L = 16, 24
    lens = rs.randint(8, L + 1, size=B).astype(np.int64)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)
    T = 5 * L
    gen = torch.Generator().manual_seed(7)
    noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=gen).to(DEV)

    def run(idx):
        out = model.duration_pitch_energy_net(None, torch.from_numpy(texts[idx]), torch.from_numpy(lens[idx]),
                                              spker_embeds=torch.from_numpy(spk[idx]), max_mel_len=T)
        mel = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], 2, noise[:, idx].contiguous())
        pcm = host.vocoder_infer(mel.transpose(1, 2).contiguous(), voc, lengths=out["mel_lens"].cpu().numpy() * cfg.hop_length)
        return mel, out["mel_lens"].cpu().numpy(), pcm

    mel, mel_len, pcm = run(np.arange(B))
    assert torch.isfinite(mel).all() and mel_len.tolist() == (lens * 5).tolist()
    assert all(p.dtype == np.int16 and p.shape[0] == n * cfg.hop_length for p, n in zip(pcm, mel_len))
    sub = np.asarray([0, 5, 11])
    mel_s, _, pcm_s = run(sub)
    assert torch.equal(mel_s, mel[sub])
    assert all(same_pcm(a, pcm[i], voc_form) for a, i in zip(pcm_s, sub))     # (Winograd form: the 16-utterance batch takes it in the C = 128 stage)


def test_denoiser_full_size_properties(conv_form):
    """cfg2 size (B=32, T=512): no cross-utterance arithmetic exists on the path, so every
    utterance's output must be bit-identical to running it alone (direct conv form; with the persistent stack's Winograd
    conv — the default — the batch takes that kernel and the lone utterance the per-layer kernels: equal within
    conftest.WINO_TOL); outputs finite; and a sub-batch spot check against the oracle."""
    host = _host()
    cfg = get_config("LJSpeech")
    sd = synth_cmtts_state_dict(cfg, seed=2)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    B, T = 32, 512
    gen = torch.Generator(device="cpu").manual_seed(1234)
    cond = torch.randn(B, T, cfg.hidden, generator=gen)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen)
    t = torch.full((B,), 1095.5)
    full = model.net(x, t, cond, None)
    torch.cuda.synchronize()
    assert torch.isfinite(full).all()
    for b in (0, 17, 31):
        one = model.net(x[b:b + 1], t[b:b + 1], cond[b:b + 1], None)
        assert same_result(one[0], full[b], conv_form), f"utterance {b} depends on its batch ({float((one[0] - full[b]).abs().max()):.2e})"
    ref = O.denoiser_forward(sd, cfg, x[:2].numpy(), t[:2].numpy(), cond[:2].numpy(), None)
    assert np.abs(_np(full[:2]) - ref).max() < 1e-3


@pytest.mark.parametrize("L,T", [(85, 512), (171, 1024)])
def test_bench_step_vs_oracle(L, T):
    """bench.py's OWN step — LJSpeech model and inputs of bench.make_inputs (B = 32 utterances of L phonemes x 6 frames, padded
    to T), default options (the conditioner factors gathered inside the F(4,3) persistent stack, the F(4,3) FFN and pitch-predictor convs),
    T = 4 sampling steps, text -> mel — against the oracle's synthesize on utterances 0 and 31 with the same noise (no arithmetic crosses
    utterances, so the oracle on two of them is the oracle on the batch): durations / mel_len / mel2ph bit-exact, energy and pitch buckets equal
    (counted), mel |d| < 1e-3 (north_star's fp32 bound; the measured value is reported).  (85, 512) = the headline shape, (171, 1024) = north_star's."""
    import bench
    host = _host()
    cfg = get_config("LJSpeech")
    sd = synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    B, n_steps = bench.BATCH, bench.N_STEPS
    rs = np.random.RandomState(0)                                  # bench.make_inputs(cfg, seed = rank 0, ...)
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    lens = np.full((B,), L, np.int64)
    noise = torch.randn(n_steps + 1, B, 1, T, cfg.n_mels, generator=torch.Generator(device="cpu").manual_seed(1234))
    out = model.duration_pitch_energy_net(None, torch.from_numpy(texts).to(DEV), torch.from_numpy(lens).to(DEV), max_mel_len=T)
    assert out["cond_factors"] is not None and out["cond_factors"].p1t is not None          # the step's default path
    mel = host.sample_with_cond(model, out["cond_ct"], None, n_steps, noise.to(DEV), factors=out["cond_factors"])
    host.synchronize()
    sel = [0, B - 1]
    ref, ref_len, st = O.synthesize(sd, cfg, texts[sel], lens[sel], None, n_steps, [noise[i][sel].numpy() for i in range(n_steps + 1)], max_mel_len=T)
    assert np.array_equal(_np(out["mel_lens"])[sel], ref_len) and (ref_len == 6 * L).all()
    assert np.array_equal(_np(out["d_rounded"])[sel], st["d_rounded"])
    Tm = min(st["mel2ph"].shape[1], T)                                # the oracle's mel2ph is as wide as the longest utterance (171 x 6 = 1026 frames are truncated to the 1024 bucket); beyond it: padding
    assert np.array_equal(_np(out["mel2ph"])[sel][:, :Tm], st["mel2ph"][:, :Tm]) and not _np(out["mel2ph"])[sel][:, Tm:].any()
    e_flips = int((_np(out["e_idx"])[sel] != st["e_idx"]).sum())
    p_flips = int((_np(out["p_predictions"]["p_idx"])[sel] != st["p_idx"]).sum())
    err = float(np.abs(_np(mel)[sel] - ref).max())
    report(f"BENCH_STEP L={L} T={T}: utterances 0 / {B - 1} vs the oracle: energy-bucket flips {e_flips}, pitch-bucket flips {p_flips} of {st['p_idx'].size}, "
           f"max|d mel| (T = {n_steps}) {err:.2e}")
    assert e_flips == 0 and p_flips == 0
    assert torch.isfinite(mel).all() and err < 1e-3


@pytest.mark.parametrize("variant,T", [("LJSpeech", 512), ("VCTK", 333), ("VCTK", 40)])
def test_fused_resblock_bitwise(variant, T):
    """The fused residual-block kernel keeps the (chunk, tap, k) accumulation order of the three-launch
    form, so the whole denoiser must agree BITWISE between the two (ragged T exercises the tile edges)."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=4))
    B = 3
    gen = torch.Generator(device="cpu").manual_seed(T)
    cond = torch.randn(B, T, cfg.hidden, generator=gen)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen)
    spk = torch.randn(B, cfg.hidden, generator=gen) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5)
    prev = lib.cmtts_set_fused_resblock(1)
    try:
        fused = model.net(x, t, cond, spk)
        lib.cmtts_set_fused_resblock(0)
        unfused = model.net(x, t, cond, spk)
    finally:
        lib.cmtts_set_fused_resblock(prev)
    torch.cuda.synchronize()
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, unfused), float((fused - unfused).abs().max())


@pytest.mark.parametrize("variant,B,T", [("LJSpeech", 2, 200), ("VCTK", 3, 64), ("VCTK", 1, 130), ("LJSpeech", 5, 1000),
                                         ("VCTK", 32, 512), ("LJSpeech", 40, 300)])
def test_persistent_denoiser_bitwise(variant, B, T, conv_form):
    """denoiser_persist.hip (all residual layers in one launch: x and skip resident in registers, edge columns
    exchanged between neighbouring tiles through tagged granules) must agree BITWISE with the per-layer kernels:
    multi-tile utterances exercise the in-kernel halo exchange, ragged T the masked tail, B=40 the utterance
    chunking (320 workgroups > 256 CUs), and the sampler runs it back to back (stale tags from the last call)."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=5))
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    cond = torch.randn(B, T, cfg.hidden, generator=gen)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen)
    spk = torch.randn(B, cfg.hidden, generator=gen) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5)
    noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    cond_ct = cond.transpose(1, 2).contiguous().to(DEV)
    spk_d = spk.to(DEV) if spk is not None else None
    prev = lib.cmtts_set_persistent_denoiser(2)
    try:
        one = model.net(x, t, cond, spk)
        mel_p = host.sample_with_cond(model, cond_ct, spk_d, 2, noise)
        lib.cmtts_set_persistent_denoiser(0)
        ref = model.net(x, t, cond, spk)
        mel_r = host.sample_with_cond(model, cond_ct, spk_d, 2, noise)
    finally:
        lib.cmtts_set_persistent_denoiser(prev)
    torch.cuda.synchronize()
    assert torch.isfinite(one).all()
    # direct form: bit for bit; Winograd form (the default): the same network within conftest.WINO_TOL
    assert same_result(one, ref, conv_form), float((one - ref).abs().max())
    assert same_result(mel_p, mel_r, conv_form), float((mel_p - mel_r).abs().max())
    if conv_form == "winograd":
        assert not torch.equal(one, ref)          # the Winograd instance did run (it is not bitwise the direct form)
        report(f"WINOGRAD {variant} B={B} T={T}: max|d| vs the per-layer kernels: one evaluation {float((one - ref).abs().max()):.2e}, "
               f"T=2 mel {float((mel_p - mel_r).abs().max()):.2e}")


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (2, 63), (1, 64), (1, 65), (3, 129)])
def test_denoiser_edge_shapes_every_mode(B, T):
    """tools/edge_check.py in the tracked suite (VERDICT r04 #5a): T = 1, 2, 63, 64, 65, 129 through every execution mode of the denoiser —
    per-layer vs forced persistent, fp32 (direct form) and bf16: bit for bit; the Winograd stack within WINO_TOL; and a one-phoneme text."""
    from conftest import WINO_TOL
    host = _host()
    lib = _lib.load()
    cfg = get_config("VCTK")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=1, dur_frames=3.0, dur_spread=0.0))
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, 1, T, 80, generator=g); cond = torch.randn(B, T, 256, generator=g); spk = torch.randn(B, 256, generator=g)
    t = torch.full((B,), 1095.5)
    outs = {}
    prev = lib.cmtts_set_persistent_denoiser(0)
    prev_w = _lib.internal_set("persist_wino", 0)
    try:
        for mode in (0, 2):
            lib.cmtts_set_persistent_denoiser(mode)
            for prec in ("fp32", "bf16"):
                model.set_precision(prec)
                outs[(mode, prec)] = model.net(x, t, cond, spk).clone()
        model.set_precision("fp32")
        wino = {}
        for wn in (1, 3):
            _lib.internal_set("persist_wino", wn)
            wino[wn] = model.net(x, t, cond, spk).clone()
    finally:
        model.set_precision("fp32")
        _lib.internal_set("persist_wino", prev_w)
        lib.cmtts_set_persistent_denoiser(prev)
    host.synchronize()
    assert all(bool(torch.isfinite(o).all()) for o in outs.values())
    assert torch.equal(outs[(0, "fp32")], outs[(2, "fp32")]) and torch.equal(outs[(0, "bf16")], outs[(2, "bf16")])
    for wn in (1, 3):
        assert float((wino[wn] - outs[(0, "fp32")]).abs().max()) <= WINO_TOL
    out = model.duration_pitch_energy_net(None, torch.tensor([[5]]), torch.tensor([1]), spker_embeds=torch.randn(1, 512))
    assert out["mel_lens"].tolist() == [3] and bool(torch.isfinite(out["cond"]).all())


@pytest.mark.parametrize("variant,B,T", [("LJSpeech", 1, 150), ("VCTK", 3, 33), ("VCTK", 2, 257), ("LJSpeech", 5, 64)])
def test_split_resblock_bitwise(variant, B, T):
    """Small batches: the residual block as two launches over four workgroups per 32-frame tile (resblock_split.hip)
    must give bit for bit what the one-workgroup kernel and the three-launch form give."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=17))
    gen = torch.Generator().manual_seed(T)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    cond = torch.randn(B, T, cfg.hidden, generator=gen).to(DEV)
    spk = torch.randn(B, cfg.hidden, generator=gen).to(DEV) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5, device=DEV)
    prev_p = lib.cmtts_set_persistent_denoiser(0)
    prev_s = lib.cmtts_set_option(b"resblock_split", 0)
    try:
        one = model.net(x, t, cond, spk).clone()
        lib.cmtts_set_option(b"resblock_split", 2)
        two = model.net(x, t, cond, spk).clone()
        lib.cmtts_set_fused_resblock(0)
        three = model.net(x, t, cond, spk)
        torch.cuda.synchronize()
    finally:
        lib.cmtts_set_fused_resblock(1)
        lib.cmtts_set_option(b"resblock_split", prev_s)
        lib.cmtts_set_persistent_denoiser(prev_p)
    assert torch.isfinite(two).all()
    assert torch.equal(two, one) and torch.equal(two, three)


def test_xres_conv_bitwise(models):
    """conv_xres.hip (k=9 FFN conv with the utterance's X tile resident in LDS, 96-column tiles) keeps the generic
    kernel's accumulation order and epilogue: the text encoder output must not change by a bit (B=32 x L=85 takes the
    X-resident path, the ragged golden batch the generic one).  Round 2: the same for the LayerNorm prologue (the fused
    normalisation keeps layernorm_ct_kernel's summation order) and for the in- / out-projections on that kernel."""
    host = _host()
    lib = _lib.load()
    g, cfg, sd, model = models("LJSpeech")
    rs = np.random.RandomState(11)
    B, L = 32, 85
    lens = rs.randint(40, L + 1, size=B).astype(np.int64)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    prev = _lib.internal_set(b"ffn_xres", 1)
    prev_t = _lib.internal_set(b"text_xres", 0)
    prev_w = _lib.internal_set(b"ffn_wino", 0)        # the direct form of the FFN conv: what these switches keep bit for bit (the F(4,3) default: test_ffn_winograd)
    try:
        run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), max_mel_len=512)
        one = run()                                   # X-resident k=9 conv, separate LayerNorm launches
        outs = []
        for bits in (9, 1, 2, 4, 7):                     # LayerNorm1 + in-projection | out-projection | LayerNorm2 + FFN conv | all
            _lib.internal_set(b"text_xres", bits)
            outs.append((f"text_xres={bits}", run()))
        _lib.internal_set(b"ffn_xres", 0)
        ref = run()
    finally:
        _lib.internal_set(b"ffn_xres", prev)
        _lib.internal_set(b"text_xres", prev_t)
        _lib.internal_set(b"ffn_wino", prev_w)
    torch.cuda.synchronize()
    for k in ("enc_out", "log_d_predictions", "cond"):
        for name, got in [("xres", one)] + outs:
            assert torch.equal(got[k], ref[k]), (name, k, float((got[k] - ref[k]).abs().max()))


def test_collated_shard_speaker_table(conv_form):
    """A ragged shard of a model whose speaker_emb is an nn.Embedding table (speaker ids as a sixth group element): the collated one-call text
    side through BucketedSynthesizer against every group alone, bit for bit."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("VCTK_table")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=14, dur_frames=4.0, dur_spread=0.0))
    rs = np.random.RandomState(3)
    n_steps, groups = 2, []
    for bucket, n in ((256, 8), (512, 6), (128, 4)):
        Lmax = bucket // 4
        ln = np.maximum((rs.uniform(0.4, 1.0, size=n) * Lmax).astype(np.int64), 1)
        ln[0] = Lmax
        tx = rs.randint(1, cfg.n_symbols, size=(n, Lmax)).astype(np.int64)
        tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
        gen = torch.Generator().manual_seed(bucket)
        groups.append((torch.from_numpy(tx).to(DEV), torch.from_numpy(ln).to(DEV), None,
                       torch.randn(n_steps + 1, n, 1, bucket, cfg.n_mels, generator=gen).to(DEV), bucket,
                       torch.from_numpy(rs.randint(0, cfg.n_speaker, size=n).astype(np.int64)).to(DEV)))
    prev = lib.cmtts_set_persistent_denoiser(2)
    try:
        alone = []
        for tx, ln, _, nz, bucket, ids in groups:
            o = model.duration_pitch_energy_net(ids, tx, ln, max_mel_len=bucket)
            alone.append((host.sample_with_cond(model, o["cond_ct"], o["speaker_emb"], n_steps, nz), o["mel_lens"]))
        host.synchronize()
    finally:
        lib.cmtts_set_persistent_denoiser(prev)
    got = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=2, trim=False).run(host.collate_groups(groups, DEV))
    old = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=2, trim=False, batch_text=False).run(groups)
    host.synchronize()
    for (m0, l0), (m1, l1), (m2, l2) in zip(alone, got, old):
        assert torch.equal(l0, l1) and torch.equal(l0, l2)
        # (the shard is too small for the persistent stack: `got` / `old` come from the per-layer kernels, `alone` from the forced stack)
        assert same_result(m0, m1, conv_form), float((m0 - m1).abs().max())
        assert same_result(m0, m2, conv_form), float((m0 - m2).abs().max())
        assert torch.equal(m1, m2)


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 1, 25), ("VCTK", 2, 85), ("LibriTTS", 3, 130), ("LJSpeech", 8, 33), ("LJSpeech", 1, 1), ("VCTK", 32, 85)])
def test_xres_small_bitwise(variant, B, L):
    """Round 4: launches that cannot fill the chip — a single request, a few utterances — take conv_xres.hip with 32-column tiles (one
    n-tile per wave: LayerNorm prologue, FFN fusion, K loop without barriers) where they took LayerNorm + the generic kernel (+ the
    FFN linear's own launch).  Same accumulation chains => the text side must not change by a bit, and a request still equals its
    row of a full batch (which takes the 96-column tiles).  The phoneme-level predictor convs likewise ("pred_xres": at every batch size —
    M = 256 never fills the chip with 96-column tiles), the previous block's LayerNorm and length mask as the prologue."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=21, dur_frames=4.0, dur_spread=0.03))
    rs = np.random.RandomState(B * 31 + L)
    lens = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=6 * L)
    prev, prev_p = _lib.internal_set(b"xres_small", 1), _lib.internal_set(b"pred_xres", 1)
    try:
        got = run()
        _lib.internal_set(b"pred_xres", 0)          # predictor convs back on the generic kernel + LayerNorm launches
        mid = run()
        _lib.internal_set(b"xres_small", 0)
        ref = run()
    finally:
        _lib.internal_set(b"xres_small", prev)
        _lib.internal_set(b"pred_xres", prev_p)
    host.synchronize()
    for k in ("enc_out", "log_d_predictions", "e_predictions", "cond_ct", "mel_lens"):
        assert torch.equal(got[k], ref[k]), (k, float((got[k].float() - ref[k].float()).abs().max()))
        assert torch.equal(mid[k], ref[k]), (k, float((mid[k].float() - ref[k].float()).abs().max()))
    # the same utterances inside a batch that fills the chip (96-column tiles, or the generic kernel for L > 96)
    rep = 40 // B + 1
    big = model.duration_pitch_energy_net(None, torch.from_numpy(np.tile(texts, (rep, 1))), torch.from_numpy(np.tile(lens, rep)),
                                          spker_embeds=None if spk is None else spk.repeat(rep, 1), max_mel_len=6 * L)
    host.synchronize()
    assert torch.equal(big["cond_ct"][:B], got["cond_ct"]) and torch.equal(big["enc_out"][:B], got["enc_out"])


@pytest.mark.parametrize("variant,B,L,T", [("LJSpeech", 32, 85, 512), ("VCTK", 3, 40, 150), ("LibriTTS", 2, 171, 1024)])
def test_cwt_in_phoneme_level_bitwise(variant, B, L, T):
    """Round 4: the pitch predictor's input projection (Linear 256 -> 128, model/modules.py:204-205) runs over the phonemes and the
    length regulator gathers its output (padding frames = the bias) instead of running over the length-regulated frames: a k = 1
    contraction commutes with the gather — every returned tensor must keep its bits (ragged lengths, truncated and padded frames)."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=22, dur_frames=5.0, dur_spread=0.03))
    rs = np.random.RandomState(L)
    lens = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=T)
    prev = _lib.internal_set(b"cwt_in_phoneme", 1)
    try:
        got = run()
        _lib.internal_set(b"cwt_in_phoneme", 0)
        ref = run()
    finally:
        _lib.internal_set(b"cwt_in_phoneme", prev)
    host.synchronize()
    assert torch.equal(got["cond_ct"], ref["cond_ct"]) and torch.equal(got["mel2ph"], ref["mel2ph"])
    for k in ("cwt", "f0_denorm", "p_idx", "f0_mean", "f0_std"):
        assert torch.equal(got["p_predictions"][k], ref["p_predictions"][k]), k


def test_ffn_fused_bitwise(models):
    """conv_xres.hip's FFN fusion (the FFN linear's K-segment partial products formed from the activated rows of the k = 9 conv
    while they are in LDS) keeps the K-segment launch's accumulation order: the text side must not change by a bit, with and
    without the LayerNorm prologue, and a single request (generic path) still equals its row of the batch."""
    host = _host()
    lib = _lib.load()
    g, cfg, sd, model = models("LJSpeech")
    rs = np.random.RandomState(12)
    B, L = 32, 85
    lens = rs.randint(40, L + 1, size=B).astype(np.int64)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    prev = _lib.internal_set(b"ffn_fused", 1)
    prev_t = _lib.internal_set(b"text_xres", 5)
    prev_w = _lib.internal_set(b"ffn_wino", 0)        # the direct form (see test_xres_conv_bitwise)
    try:
        run = lambda: model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), max_mel_len=512)
        outs = [("fused", run())]
        _lib.internal_set(b"text_xres", 0)
        outs.append(("fused, separate LayerNorm", run()))
        _lib.internal_set(b"ffn_fused", 0)
        ref = run()
        one = model.duration_pitch_energy_net(None, torch.from_numpy(texts[:1]), torch.from_numpy(lens[:1]), max_mel_len=512)
    finally:
        _lib.internal_set(b"ffn_fused", prev)
        _lib.internal_set(b"text_xres", prev_t)
        _lib.internal_set(b"ffn_wino", prev_w)
    torch.cuda.synchronize()
    for k in ("enc_out", "log_d_predictions", "cond"):
        for name, got in outs:
            assert torch.equal(got[k], ref[k]), (name, k, float((got[k] - ref[k]).abs().max()))
    assert torch.equal(one["enc_out"][0], outs[0][1]["enc_out"][0])


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 16, 100), ("LJSpeech", 5, 128), ("LibriTTS", 3, 300)])
def test_ffn_winograd(variant, B, L, form):
    """The FFT blocks' k = 9 FFN conv as three Winograd tap groups inside the fused launch (conv_xres.hip, WQ instances; fp32 models) — form 1: F(2,3) over output pairs
    (the default since round 6: three full n-tiles of pair lanes per 96-column tile, 3 VALU operations per transformed n-tile), form 2: F(4,3) over output quads (round 5) —
    against the direct form: encoder output by fp32 rounding only, durations and mel lengths equal.  And the property the direct form had by construction: ONE form at every
    shape — the 96-column tiles of a chip-filling batch, the 32-column tiles of a single request, lengths at which the direct form takes the generic kernel — so an
    utterance's encoder output does not depend on the batch it is in (bit for bit)."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=9, dur_frames=4.0, dur_spread=0.0))
    rs = np.random.RandomState(B * 7 + L)
    lens = rs.randint(max(L // 2, 1), L + 1, size=B).astype(np.int64)
    lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal((B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    tx, ln = torch.from_numpy(texts), torch.from_numpy(lens)
    run = lambda nb: model.duration_pitch_energy_net(None, tx[:nb], ln[:nb], spker_embeds=None if spk is None else spk[:nb])
    prev = _lib.internal_set(b"ffn_wino", form)
    try:
        got, one = run(B), run(1)
        _lib.internal_set(b"ffn_wino", 0)
        ref = run(B)
    finally:
        _lib.internal_set(b"ffn_wino", prev)
    torch.cuda.synchronize()
    d = float((got["enc_out"] - ref["enc_out"]).abs().max())
    report(f"FFN_WINOGRAD {'F(2,3)' if form == 1 else 'F(4,3)'} {variant} B={B} L={L}: max|d enc_out| vs the direct form {d:.2e} (scale {float(ref['enc_out'].abs().max()):.2f}); "
           f"log-durations {float((got['log_d_predictions'] - ref['log_d_predictions']).abs().max()):.2e}")
    assert torch.isfinite(got["enc_out"]).all() and 0 < d <= 2e-5
    assert torch.equal(got["mel_lens"], ref["mel_lens"]) and torch.equal(got["mel2ph"], ref["mel2ph"])
    assert torch.equal(one["enc_out"][0], got["enc_out"][0]) and torch.equal(one["log_d_predictions"][0], got["log_d_predictions"][0])


@pytest.mark.parametrize("variant,B,T", [("LJSpeech", 3, 200), ("VCTK", 2, 77), ("LJSpeech", 32, 512)])
def test_cond_gemm_bitwise(variant, B, T):
    """cond_gemm.hip (conditioner projections of all layers, X tile resident in LDS) keeps the generic kernel's
    accumulation order: the denoiser output must not change by a bit."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=7))
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    cond = torch.randn(B, T, cfg.hidden, generator=gen)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen)
    spk = torch.randn(B, cfg.hidden, generator=gen) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5)
    prev = _lib.internal_set(b"cond_gemm", 2)          # 2 = take the kernel at every size (1 leaves small batches to the generic kernel)
    try:
        one = model.net(x, t, cond, spk)
        _lib.internal_set(b"cond_gemm", 0)
        ref = model.net(x, t, cond, spk)
    finally:
        _lib.internal_set(b"cond_gemm", prev)
    torch.cuda.synchronize()
    assert lib.cmtts_set_option(b"no_such_option", 0) < 0
    assert torch.isfinite(one).all()
    assert torch.equal(one, ref), float((one - ref).abs().max())


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "fp16x3"])
@pytest.mark.parametrize("variant,B,T,layers", [("LJSpeech", 3, 200, 20), ("VCTK", 2, 77, 20), ("LJSpeech", 32, 512, 20), ("LJSpeech", 2, 200, 1)])
def test_cond_projections_operands(variant, B, T, layers, dtype):
    """The stacked conditioner projections on their own (csrc/internal_hooks.h: cmtts_internal_cond_projections): fp32 models multiply fp32
    operands (cond_gemm.hip / the generic kernel), bf16 / fp16 models operands rounded to 16 bits and fp16x3 models (hi, lo) fp16 pairs, with fp32
    accumulation (cond_gemm16.hip, round 3) at EVERY shape — the numerics of a 16-bit model must not depend on the batch.  Against float64 products of the same operands:
    what is left is fp32 accumulation error.  T = 77 / 200: ragged last tile; one layer: waves without rows in the only pass."""
    import ctypes as C
    import dataclasses
    host = _host()
    cfg = dataclasses.replace(get_config(variant), res_layers=layers)
    sd = synth_cmtts_state_dict(cfg, seed=7)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    model.set_precision(dtype)
    raw = C.CDLL(_lib.LIB_PATH)
    raw.cmtts_internal_cond_projections.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    NL, Cc, H = cfg.res_layers, cfg.res_channels, cfg.hidden
    W = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.weight"]))[:, :, 0] for l in range(NL)], 0)
    bias = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.bias"])) for l in range(NL)], 0)
    cond = torch.randn(B, H, T, generator=torch.Generator().manual_seed(B * 100 + T))
    cd = cond.to(DEV)
    cp = torch.full((B, NL * Cc, T), float("nan"), device=DEV)
    assert raw.cmtts_internal_cond_projections(model._h, C.c_void_p(cd.data_ptr()), B, T, C.c_void_p(cp.data_ptr()), None) == 0
    torch.cuda.synchronize()
    def q(a):                  # the operand the device multiplies, as a float64 tensor
        if dtype == "fp16x3":  # hi + lo fp16 pair (the dropped lo * lo term is 2^-22 relative)
            hi = a.half().float()
            return hi.double() + (a - hi).half().double()
        return a.to({"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]).double()
    ref = torch.einsum("mk,bkt->bmt", q(W), q(cond)) + bias.double()[None, :, None]
    got = cp.cpu().double()
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-6 * scale, (float((got - ref).abs().max()), scale)
    if dtype in ("bf16", "fp16"):      # and the operands really were rounded: far from the fp32-operand products
        ref32 = torch.einsum("mk,bkt->bmt", W.double(), cond.double()) + bias.double()[None, :, None]
        assert float((got - ref32).abs().max()) > 50e-6 * scale
    model.set_precision("fp32")


def test_model_without_pitch_table_factor_takes_the_dense_gemm():
    """a model whose stacked conditioner projection has no pitch-table factor (res_layers = 1: NL * C is not a
    multiple of 512) must go through DurationPitchSpeakerNet.forward and the sampler exactly as before the factors existed — the
    frame-level call skips the phoneme-level factor, CondFactors.usable() is false, the dense GEMM runs — instead of failing with
    CMTTS_E_UNSUPPORTED."""
    import dataclasses
    host = _host()
    cfg = dataclasses.replace(get_config("VCTK"), res_layers=1)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=23, dur_frames=5.8, dur_spread=0.03))
    rs = np.random.RandomState(3)
    B, L, T = 3, 30, 192
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    lens = np.array([L, 21, 9], np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32))
    out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=T)
    nz = torch.randn(3, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(1)).to(DEV)
    f = out.get("cond_factors")
    mel = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], 2, nz, factors=f)
    ref = host.sample_with_cond(model, out["cond_ct"].clone(), out["speaker_emb"], 2, nz)      # a copy carries no factors: the dense GEMM
    host.synchronize()
    assert torch.isfinite(mel).all() and torch.equal(mel, ref)


@pytest.mark.parametrize("variant,B,T", [("VCTK", 33, 513), ("LJSpeech", 5, 65), ("VCTK", 1, 5000), ("LJSpeech", 70, 300), ("VCTK", 7, 1),
                                         ("LJSpeech", 3, 63), ("VCTK", 32, 512)])
def test_winograd_stack_odd_shapes(variant, B, T):
    """VERDICT r04 #5a: the Winograd forms of the fp32 persistent stack work on frame PAIRS (F(2,3)) or QUADS (F(4,3), the default since
    round 5) — odd T, a one-frame utterance, a lone tail tile, utterance chunking (70 x 5 tiles > 256 CUs) and 79-tile utterances are
    where a tile-wise transform breaks.  Three stacks on the same inputs: direct (bitwise the per-layer kernels, proven elsewhere), the
    F(2,3) instances and the F(4,3) instances (persist_wino = 2, round 5's one-wave-per-SIMD F(2,3) stack, left the build in round 6 —
    tools/attic/ — and runs as 1): every Winograd stack within WINO_TOL of the direct form — one network evaluation and a T = 2 sample."""
    from conftest import WINO_TOL
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
    g = torch.Generator().manual_seed(B * 7 + T)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=g).to(DEV)
    cond = torch.randn(B, T, cfg.hidden, generator=g).to(DEV)
    spk = torch.randn(B, cfg.hidden, generator=g).to(DEV) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5, device=DEV)
    noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=g).to(DEV)
    cond_ct = cond.transpose(1, 2).contiguous()
    outs, mels = {}, {}
    prev = lib.cmtts_set_persistent_denoiser(2)
    prev_w = _lib.internal_set("persist_wino", 0)
    try:
        for wn in (0, 1, 2, 3):
            _lib.internal_set("persist_wino", wn)
            outs[wn] = model.net(x, t, cond, spk).clone()
            mels[wn] = host.sample_with_cond(model, cond_ct, spk, 2, noise).clone()
        # the per-model option on top of the process default (3): 2 = F(2,3), 0 = direct
        try:
            model.set_option("winograd", 2)
            opt2 = model.net(x, t, cond, spk).clone()
            model.set_option("winograd", 0)
            opt0 = model.net(x, t, cond, spk).clone()
        finally:
            model.set_option("winograd", 1)
    finally:
        _lib.internal_set("persist_wino", prev_w)
        lib.cmtts_set_persistent_denoiser(prev)
    host.synchronize()
    for wn in (1, 3):
        assert torch.isfinite(outs[wn]).all() and torch.isfinite(mels[wn]).all()
    assert torch.equal(outs[1], outs[2]) and torch.equal(mels[1], mels[2]), float((outs[1] - outs[2]).abs().max())
    assert torch.equal(opt2, outs[1]) and torch.equal(opt0, outs[0])
    d1, dm = float((outs[1] - outs[0]).abs().max()), float((mels[1] - mels[0]).abs().max())
    d3, dm3 = float((outs[3] - outs[0]).abs().max()), float((mels[3] - mels[0]).abs().max())
    report(f"WINOGRAD_ODD {variant} B={B} T={T}: max|d| vs the direct stack: one evaluation F(2,3) {d1:.2e} F(4,3) {d3:.2e}, T=2 mel {dm:.2e} / {dm3:.2e}; "
           "8-wave == one-wave-per-SIMD bitwise")
    assert 0 < d1 <= WINO_TOL and dm <= WINO_TOL, (d1, dm)
    assert 0 < d3 <= WINO_TOL and dm3 <= WINO_TOL, (d3, dm3)


@pytest.mark.parametrize("variant,B,L,T", [("LJSpeech", 32, 85, 512), ("VCTK", 3, 40, 200), ("LibriTTS", 2, 171, 1024), ("LJSpeech", 1, 5, 33)])
def test_cond_factored(variant, B, L, T, conv_form):
    """Round 4: the conditioner projections expanded from their factors — cp[:, t] = (Wc out1)[:, mel2ph[t] - 1] + (Wc pitch_embed^T + b)[:, p_idx[t]]
    (cmtts_frame_forward_sub's cond_p1, cond_expand_kernel) — against (a) the float64 product Wc cond + b of the conditioning the same call
    returned: the same bound as the dense GEMM's (fp32 accumulation error only), and (b) the dense path on the device.  Then the sampler: mel with
    and without the factors within 2e-5 (W (a + b) and W a + W b round differently), durations / buckets untouched (they are upstream)."""
    import ctypes as C
    host = _host()
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=19, dur_frames=float(max(1, T // L)) - 0.2, dur_spread=0.03)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(sd)
    rs = np.random.RandomState(B * 7 + L)
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    lens = np.maximum((rs.uniform(0.5, 1.0, size=B) * L).astype(np.int64), 1)
    lens[0] = L
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    out = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=T)
    f = out["cond_factors"]
    assert f is not None and f.matches(out["cond_ct"])
    raw = C.CDLL(_lib.LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    raw.cmtts_internal_cond_projections.argtypes = [vp, vp, ci, ci, vp, vp]
    raw.cmtts_internal_cond_factored.argtypes = [vp, vp, ci, ci, vp, vp, ci, ci, vp, vp]
    NL, Cc = cfg.res_layers, cfg.res_channels
    cp_d = torch.full((B, NL * Cc, T), float("nan"), device=DEV)
    cp_f = torch.full((B, NL * Cc, T), float("nan"), device=DEV)
    assert raw.cmtts_internal_cond_projections(model._h, vp(out["cond_ct"].data_ptr()), B, T, vp(cp_d.data_ptr()), None) == 0
    assert raw.cmtts_internal_cond_factored(model._h, vp(f.p1.data_ptr()), f.p1_ld, f.L, vp(f.mel2ph.data_ptr()), vp(f.p_idx.data_ptr()), B, T,
                                            vp(cp_f.data_ptr()), None) == 0
    torch.cuda.synchronize()
    W = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.weight"]))[:, :, 0] for l in range(NL)], 0)
    bias = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.bias"])) for l in range(NL)], 0)
    ref = torch.einsum("mk,bkt->bmt", W.double(), out["cond_ct"].cpu().double()) + bias.double()[None, :, None]
    scale = float(ref.abs().max())
    e_d, e_f = float((cp_d.cpu().double() - ref).abs().max()), float((cp_f.cpu().double() - ref).abs().max())
    assert torch.isfinite(cp_f).all()
    assert e_d <= 2e-6 * scale and e_f <= 3e-6 * scale, (e_d, e_f, scale)
    assert not torch.equal(cp_d, cp_f)          # another association, not another result
    # the sampler with and without the factors
    gen = torch.Generator().manual_seed(5)
    nz = torch.randn(5, B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    for n_steps in (1, 4):
        m_f = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], n_steps, nz, factors=f)
        dense_ct = out["cond_ct"].clone()              # a copy carries no factors: the dense GEMM
        m_d = host.sample_with_cond(model, dense_ct, out["speaker_emb"], n_steps, nz)
        prev = _lib.internal_set("cond_factored", 0)
        try:
            m_off = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], n_steps, nz, factors=f)
        finally:
            _lib.internal_set("cond_factored", prev)
        # the persistent kernel gathering the factors itself (FACT instances, no cp tensor) against expanding them first: the same bits
        # (forced persistent so that the small shapes take that kernel too)
        lib = _lib.load()
        prev_p = lib.cmtts_set_persistent_denoiser(2)
        prev_k = _lib.internal_set("cond_inkernel", 1)
        try:
            m_ik = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], n_steps, nz, factors=f)
            _lib.internal_set("cond_inkernel", 0)
            m_ex = host.sample_with_cond(model, out["cond_ct"], out["speaker_emb"], n_steps, nz, factors=f)
        finally:
            _lib.internal_set("cond_inkernel", prev_k)
            lib.cmtts_set_persistent_denoiser(prev_p)
        host.synchronize()
        assert torch.equal(m_ik, m_ex), float((m_ik - m_ex).abs().max())
        # the large shape takes the persistent stack by itself (m_f: the same kernel); at the small shapes m_f came from the per-layer
        # kernels: bitwise the direct form, within WINO_TOL of the (default) Winograd form
        assert same_result(m_ik, m_f, conv_form, strict=B * ((T + 63) // 64) * 2 > 256), float((m_ik - m_f).abs().max())
        assert torch.equal(m_off, m_d)                                     # the switch restores the dense GEMM
        err = float((m_f - m_d).abs().max())
        report(f"COND_FACTORED {variant} B={B} T={T} steps={n_steps}: cp max|d| vs f64 dense {e_d:.2e} factored {e_f:.2e} (scale {scale:.2f}); max|dmel| factored vs dense {err:.2e}")
        assert err < 2e-5, err
    # a modified conditioning tensor no longer matches its factors: the dense GEMM runs on what the caller passes
    out["cond_ct"].add_(0.0)
    assert not f.matches(out["cond_ct"])


def test_ragged_text_batch_bitwise():
    """Round 4 (third bullet): the phoneme-level half of all bucket groups of a shard in ONE call
    (cmtts_text_forward_ragged, per-utterance pad_lens) + cmtts_frame_forward_sub per group: every group's conditioning, durations,
    buckets and factors are bit-identical to running the group alone (where the padded length enters the reference's arithmetic — the
    speaker vector added to every padded column, the unmasked energy predictor — the kernels stop at the group's own length)."""
    host = _host()
    lib = _lib.load()
    import ctypes as C
    for variant in ("LibriTTS", "LJSpeech"):
        cfg = get_config(variant)
        model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=12, dur_frames=4.0, dur_spread=0.03))
        rs = np.random.RandomState(16)
        groups = []
        for bucket, n in ((128, 5), (512, 3), (256, 8), (1024, 2), (384, 1)):      # L = 32 / 128 / 64 / 256 (the long-attention class) / 96
            Lmax = bucket // 4
            ln = np.maximum((rs.uniform(0.3, 1.0, size=n) * Lmax).astype(np.int64), 1)
            ln[0] = Lmax                                                       # one utterance fills its group: its last column has no neighbour
            tx = rs.randint(1, cfg.n_symbols, size=(n, Lmax)).astype(np.int64)
            tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
            sp = torch.from_numpy(rs.standard_normal(size=(n, cfg.external_speaker_dim)).astype(np.float32)).to(DEV) if cfg.multi_speaker else None
            groups.append((torch.from_numpy(tx).to(DEV), torch.from_numpy(ln).to(DEV), sp, None, bucket))
        coll = host.collate_groups(groups, DEV)
        assert len(coll.batches) == 2                                          # L <= 192 and L > 192 stay apart (different attention kernels)
        for k, tb in enumerate(coll.batches):
            B, L = tb["texts"].shape
            f32 = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=DEV)
            log_d, d_r, e_p = f32(B, L), f32(B, L), f32(B, L)
            e_i = torch.zeros(B, L, dtype=torch.int64, device=DEV)
            mel_len = torch.zeros(B, dtype=torch.int64, device=DEV)
            spk_o = f32(B, cfg.hidden) if cfg.multi_speaker else None
            nb = lib.cmtts_text_workspace_bytes(model._h, B, L)
            tws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            _lib.check(lib.cmtts_text_forward_ragged(model._h, p(tb["texts"]), p(tb["src_lens"]), p(tb["pad_lens"]), p(tb["spk"]), None, B, L, 1.0,
                                                     p(log_d), p(d_r), p(mel_len), p(e_p), p(e_i), None, p(spk_o), p(tws), nb, None))
            for (i, b0, n, Lg) in tb["members"]:
                tx, ln, sp, _, bucket = groups[i]
                alone = model.duration_pitch_energy_net(None, tx, ln, spker_embeds=sp, max_mel_len=bucket)
                cond_ct, fac = host._frame_forward_sub(model, tws, B, L, b0, n, bucket, ("frame_test", i))
                host.synchronize()
                sl = slice(b0, b0 + n)
                assert torch.equal(mel_len[sl], alone["mel_lens"]), (variant, i)
                assert torch.equal(d_r[sl, :Lg], alone["d_rounded"]) and torch.equal(log_d[sl, :Lg], alone["log_d_predictions"]), (variant, i)
                assert torch.equal(e_p[sl, :Lg], alone["e_predictions"]) and torch.equal(e_i[sl, :Lg], alone["e_idx"]), (variant, i)
                assert torch.equal(cond_ct, alone["cond_ct"]), (variant, i, float((cond_ct - alone["cond_ct"]).abs().max()))
                fa = alone["cond_factors"]
                assert torch.equal(fac.mel2ph, fa.mel2ph) and torch.equal(fac.p_idx, fa.p_idx)
                assert torch.equal(fac.p1[:, :, :Lg], fa.p1[:, :, :Lg]), (variant, i)
                if cfg.multi_speaker:
                    assert torch.equal(spk_o[sl], alone["speaker_emb"])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 3, 85), ("VCTK", 2, 31), ("LibriTTS", 2, 170), ("LJSpeech", 1, 1)])
def test_text16_xresident_bitwise(variant, B, L, dtype):
    """conv_xt16.hip (the text16 convs with 256 input channels: whole x^T tile of an utterance staged once, hand-issued weight ring, no barrier
    in the K loop) keeps conv_mfma16.hip's conversions, (chunk, tap, k-group) order and epilogue: every output of the text side — floats and
    integers — must not change by a bit.  L = 85 / 31 / 1: one ragged 96-column tile; 170: two tiles; ragged lengths exercise the masks."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=31, dur_frames=4.0, dur_spread=0.03))
    rs = np.random.RandomState(L)
    ln = np.maximum((rs.uniform(0.4, 1.0, size=B) * L).astype(np.int64), 1); ln[0] = L
    tx = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    tx[np.arange(L)[None, :] >= ln[:, None]] = 0
    spk = torch.randn(B, cfg.external_speaker_dim, generator=torch.Generator().manual_seed(L)) if cfg.multi_speaker else None
    model.set_precision(dtype)
    model.set_option("text16", 1)
    keys = ("enc_out", "log_d_predictions", "e_predictions", "cond", "d_rounded", "mel_lens", "mel2ph")
    prev = _lib.internal_set(b"text_xt16", 0)
    try:
        ref = model.duration_pitch_energy_net(None, torch.from_numpy(tx), torch.from_numpy(ln), spker_embeds=spk)
        ref = {k: ref[k].clone() for k in keys} | {"cwt": ref["p_predictions"]["cwt"].clone()}
        _lib.internal_set(b"text_xt16", 1)
        got = model.duration_pitch_energy_net(None, torch.from_numpy(tx), torch.from_numpy(ln), spker_embeds=spk)
        got = {k: got[k].clone() for k in keys} | {"cwt": got["p_predictions"]["cwt"].clone()}
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"text_xt16", prev)
        model.set_option("text16", 0)
        model.set_precision("fp32")
    for k in ref:
        assert torch.isfinite(got[k].float()).all(), k
        assert torch.equal(got[k], ref[k]), (k, float((got[k].float() - ref[k].float()).abs().max()))


def test_bucketed_synthesizer_streams_match_sequential():
    """configs[3] shape: bucket groups on separate HIP streams (own workspaces) must give exactly the results of running
    the groups one after the other."""
    host = _host()
    cfg = get_config("LibriTTS")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=12, dur_frames=4.0, dur_spread=0.0))
    rs = np.random.RandomState(5)
    groups = []
    for bucket in (64, 128, 256, 192):
        n, Lmax = 3, bucket // 4
        ln = np.maximum((rs.uniform(0.5, 1.0, size=n) * Lmax).astype(np.int64), 1)
        ln[0] = Lmax
        tx = rs.randint(1, cfg.n_symbols, size=(n, Lmax)).astype(np.int64)
        tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
        gen = torch.Generator().manual_seed(bucket)
        groups.append((torch.from_numpy(tx).to(DEV), torch.from_numpy(ln).to(DEV),
                       torch.randn(n, cfg.external_speaker_dim, generator=gen).to(DEV),
                       torch.randn(3, n, 1, bucket, cfg.n_mels, generator=gen).to(DEV), bucket))
    seq = []
    for tx, ln, spk, nz, bucket in groups:
        o = model.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=bucket)
        seq.append((host.sample_with_cond(model, o["cond_ct"], o["speaker_emb"], 2, nz), o["mel_lens"]))
    torch.cuda.synchronize()
    par = host.BucketedSynthesizer(model, n_steps=2, n_streams=3, mode="streams").run(groups)
    torch.cuda.synchronize()
    for (m0, l0), (m1, l1) in zip(seq, par):
        assert torch.equal(l0, l1) and torch.equal(m0, m1)
    # a 16-bit model: the default ("ragged") mode keeps one stream per group (the one-launch form is the fp32 persistent kernel's)
    model.set_precision("bf16")
    try:
        seq16 = []
        for tx, ln, spk, nz, bucket in groups:
            o = model.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=bucket)
            seq16.append(host.sample_with_cond(model, o["cond_ct"], o["speaker_emb"], 2, nz))
        par16 = host.BucketedSynthesizer(model, n_steps=2, n_streams=3).run(groups)
        host.synchronize()
    finally:
        model.set_precision("fp32")
    for m0, (m1, _) in zip(seq16, par16):
        assert torch.equal(m0, m1)


@pytest.mark.parametrize("n_steps", [1, 4])
def test_ragged_one_launch_shard_bitwise(n_steps, conv_form):
    """VERDICT r02 next #2, BASELINE.json configs[3]: all bucket groups of a ragged shard through ONE persistent launch per
    evaluation (cmtts_sample_ragged, tile-descriptor list).  (a) untrimmed: every frame of every padded group is bit-identical
    to running the group alone (parity is defined per padded bucket, model/modules.py:429-430); (b) trimmed to mel_len + 16
    frames (+ the sampler's receptive field): every frame below mel_len + 16 is still bit-identical (direct, F(2,3); within
    WINO_TRIM_TOL in the F(4,3) form: conftest.py), frames beyond the computed range are zeros; (c) a shard with more active tiles than CUs runs in rounds and stays bit-identical."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("LibriTTS")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=12, dur_frames=4.0, dur_spread=0.0))
    rs = np.random.RandomState(6)

    def make(buckets, n):
        groups = []
        for bucket in buckets:
            Lmax = bucket // 4
            ln = np.maximum((rs.uniform(0.3, 1.0, size=n) * Lmax).astype(np.int64), 1)
            ln[0] = Lmax
            tx = rs.randint(1, cfg.n_symbols, size=(n, Lmax)).astype(np.int64)
            tx[np.arange(Lmax)[None, :] >= ln[:, None]] = 0
            gen = torch.Generator().manual_seed(bucket + n)
            groups.append((torch.from_numpy(tx).to(DEV), torch.from_numpy(ln).to(DEV),
                           torch.randn(n, cfg.external_speaker_dim, generator=gen).to(DEV),
                           torch.randn(n_steps + 1, n, 1, bucket, cfg.n_mels, generator=gen).to(DEV), bucket))
        return groups

    def alone(groups):
        prev = lib.cmtts_set_persistent_denoiser(2)       # the same persistent kernel, one uniform launch per group
        try:
            seq = []
            for tx, ln, spk, nz, bucket in groups:
                o = model.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=bucket)
                seq.append((host.sample_with_cond(model, o["cond_ct"], o["speaker_emb"], n_steps, nz, factors=o["cond_factors"]), o["mel_lens"]))
            host.synchronize()
            return seq
        finally:
            lib.cmtts_set_persistent_denoiser(prev)

    groups = make((128, 256, 512, 384), 8)                 # 8 * (2 + 4 + 8 + 6) = 160 padded tiles: enough for the one-launch form
    seq = alone(groups)
    full = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=3, trim=False, batch_text=True).run(groups)      # one-call text side
    trim = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=3, tail_frames=16).run(groups)                  # one text side per group
    host.synchronize()
    saved = 0
    for (m0, l0), (m1, l1), (m2, l2) in zip(seq, full, trim):
        assert torch.equal(l0, l1) and torch.equal(l0, l2)
        assert torch.equal(m0, m1), float((m0 - m1).abs().max())
        for b, n in enumerate(l0.tolist()):
            keep = min(n + 16, m0.shape[1])
            assert same_trimmed(m2[b, :keep], m0[b, :keep], conv_form), (b, n, float((m2[b, :keep] - m0[b, :keep]).abs().max()))
            cut = min(m0.shape[1], (n + 16 + cfg.res_layers + 63) // 64 * 64)      # the last evaluation's computed range
            assert not m2[b, cut:].any()
            saved += m0.shape[1] - cut
    assert saved > 0, "the shard has nothing to trim: the test is vacuous"
    if n_steps == 1:
        # tail_frames = 16 covers HiFi-GAN's receptive field (conv_pre 3 frames + the k = 11 ResBlock of the first stage 7.5 + ...): the
        # int16 wav of every utterance, cut to mel_len * hop as vocoder_infer does, is the same from the trimmed and the full mel
        hcfg = HifiGanConfig()
        voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=3))
        for (m0, l0), (m2, _) in zip(seq[:2], trim[:2]):
            lens_w = (l0 * cfg.hop_length).tolist()
            w_full = host.vocoder_infer(m0.transpose(1, 2).contiguous(), voc, lengths=lens_w)
            w_trim = host.vocoder_infer(m2.transpose(1, 2).contiguous(), voc, lengths=lens_w)
            if trim_exact(conv_form):
                assert all(np.array_equal(a, b) for a, b in zip(w_full, w_trim))
            else:      # F(4,3): mels within WINO_TRIM_TOL -> samples within one int16 step
                assert all(a.shape == b.shape and int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()) <= 1 for a, b in zip(w_full, w_trim))
    # (d) 15 x 16 + 8 x 4 = 272 padded tiles: leaving the small group out fits one round — it is set aside for the ordinary sampler
    # (per-layer kernels on the side stream) while the large one takes the persistent launch
    # (the set-aside group: within WINO_TOL of `alone` in the Winograd form, the persistent group bit for bit in either form)
    mixed = make((1024,), 15) + make((256,), 8)
    seq = alone(mixed)
    got = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=2, trim=False, batch_text=True).run(mixed)
    host.synchronize()
    for gi, ((m0, l0), (m1, l1)) in enumerate(zip(seq, got)):
        assert torch.equal(l0, l1) and same_result(m0, m1, conv_form, strict=gi == 0), (gi, float((m0 - m1).abs().max()))
    # (d') ADVICE r03: more groups than streams, trimmed — the late (large) group shares the ONE stream with the early (set-aside)
    # group; main must be ordered behind the late group's text side by its own event before it reads mel_lens / cond
    got = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=1, tail_frames=16).run(mixed)
    host.synchronize()
    for gi, ((m0, l0), (m1, l1)) in enumerate(zip(seq, got)):
        assert torch.equal(l0, l1)
        for b, n in enumerate(l0.tolist()):
            keep = min(n + 16, m0.shape[1])
            assert (same_trimmed(m1[b, :keep], m0[b, :keep], conv_form) if gi == 0 else same_result(m1[b, :keep], m0[b, :keep], conv_form)), \
                (gi, b, n, float((m1[b, :keep] - m0[b, :keep]).abs().max()))
    big = make((512, 1024), 14)                            # 14 * (8 + 16) = 336 padded tiles > 256 CUs: rounds of whole utterances
    seq = alone(big)
    got = host.BucketedSynthesizer(model, n_steps=n_steps, n_streams=2, trim=False).run(big)
    host.synchronize()
    for (m0, l0), (m1, l1) in zip(seq, got):
        assert torch.equal(l0, l1) and torch.equal(m0, m1), float((m0 - m1).abs().max())


def test_two_persistent_launches_on_two_streams(conv_form):
    """Two persistent denoiser launches issued on different streams are chained by the library (each needs all of its
    workgroups resident): both finish, nothing times out, results are those of the per-layer kernels."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    m1 = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=13))
    m2 = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=14))
    B, T = 24, 512
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    cond = torch.randn(B, T, cfg.hidden, generator=gen).to(DEV)
    t = torch.full((B,), 1095.5, device=DEV)
    prev = lib.cmtts_set_persistent_denoiser(0)
    try:
        r1, r2 = m1.net(x, t, cond, None), m2.net(x, t, cond, None)
        torch.cuda.synchronize()
        lib.cmtts_set_persistent_denoiser(2)
        s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
        for _ in range(3):
            with torch.cuda.stream(s1):
                o1 = m1.net(x, t, cond, None)
            with torch.cuda.stream(s2):
                o2 = m2.net(x, t, cond, None)
        torch.cuda.synchronize()
        assert same_result(o1, r1, conv_form) and same_result(o2, r2, conv_form)
        q1, q2 = m1.net(x, t, cond, None), m2.net(x, t, cond, None)          # would raise if a neighbour wait had timed out
        torch.cuda.synchronize()
        assert torch.equal(o1, q1) and torch.equal(o2, q2)                   # the stack alone on one stream: the same bits in either form
    finally:
        lib.cmtts_set_persistent_denoiser(prev)


def test_small_persistent_launches_share_the_chip(conv_form):
    """Persistent launches are admitted by capacity: grids of different streams that fit the CU count together run
    side by side (here 3 models x 48 workgroups, then a 4th stream with 192 that must wait for some of them);
    every result equals the per-layer kernels' and nothing times out."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    shapes = [(6, 512), (6, 512), (6, 500), (12, 1024)]
    models = [host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=30 + i)) for i in range(4)]
    gen = torch.Generator().manual_seed(5)
    data = [(torch.randn(B, 1, T, cfg.n_mels, generator=gen).to(DEV), torch.randn(B, T, cfg.hidden, generator=gen).to(DEV),
             torch.full((B,), 1095.5, device=DEV)) for B, T in shapes]
    prev = lib.cmtts_set_persistent_denoiser(0)
    try:
        ref = [m.net(x, t, c, None) for m, (x, c, t) in zip(models, data)]
        torch.cuda.synchronize()
        lib.cmtts_set_persistent_denoiser(2)
        streams = [torch.cuda.Stream(device=DEV) for _ in range(4)]
        out = [None] * 4
        for _ in range(3):
            for i, (m, (x, c, t)) in enumerate(zip(models, data)):
                with torch.cuda.stream(streams[i]):
                    out[i] = m.net(x, t, c, None)
        torch.cuda.synchronize()
        for o, r in zip(out, ref):
            assert same_result(o, r, conv_form)
        quiet = [m.net(x, t, c, None) for m, (x, c, t) in zip(models, data)]      # one stream, nothing else running: the same bits in either form
        torch.cuda.synchronize()
        for o, q in zip(out, quiet):
            assert torch.equal(o, q)
        models[0].net(*[data[0][j] for j in (0, 2, 1)], None)          # would raise if a neighbour wait had timed out
    finally:
        lib.cmtts_set_persistent_denoiser(prev)


def test_persistent_denoiser_under_uneven_load(conv_form):
    """The in-kernel edge-column hand-off must not depend on the workgroups starting together: run the persistent stack
    while other streams keep part of the GPU busy (its workgroups then become resident at different times and wait for
    each other through the tagged granules), several times, and compare bitwise with the per-layer result."""
    host = _host()
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=9))
    B, T = 32, 512
    gen = torch.Generator(device="cpu").manual_seed(99)
    cond = torch.randn(B, T, cfg.hidden, generator=gen).to(DEV)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen).to(DEV)
    t = torch.full((B,), 1095.5, device=DEV)
    prev = lib.cmtts_set_persistent_denoiser(0)
    try:
        ref = model.net(x, t, cond, None)
        torch.cuda.synchronize()
        lib.cmtts_set_persistent_denoiser(2)
        quiet = model.net(x, t, cond, None)        # the stack on an idle GPU
        torch.cuda.synchronize()
        assert same_result(quiet, ref, conv_form), float((quiet - ref).abs().max())
        side = [torch.cuda.Stream(device=DEV) for _ in range(2)]
        a = torch.randn(4096, 4096, device=DEV)
        for rep in range(4):
            for i, s in enumerate(side):            # long GEMMs / elementwise chains on other streams
                with torch.cuda.stream(s):
                    for _ in range(3 + 2 * rep + i):
                        b = a @ a
                        b = torch.tanh(b * 1e-3)
            out = model.net(x, t, cond, None)
            torch.cuda.synchronize()
            assert torch.equal(out, quiet), (rep, float((out - quiet).abs().max()))
    finally:
        lib.cmtts_set_persistent_denoiser(prev)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("variant,B,T", [("VCTK", 2, 200), ("LJSpeech", 32, 512), ("VCTK", 40, 300), ("LJSpeech", 3, 129), ("LJSpeech", 70, 500)])
def test_persistent_denoiser_lp_bitwise(variant, B, T, dtype):
    """denoiser_persist_lp.hip (16-bit MFMA operands, persistent stack) must agree BITWISE with the per-layer 16-bit
    kernels (resblock_fused_lp.hip): same conversions, same (tap, k-group) accumulation order."""
    host = _host()
    lib = _lib.load()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=6))
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    cond = torch.randn(B, T, cfg.hidden, generator=gen)
    x = torch.randn(B, 1, T, cfg.n_mels, generator=gen)
    spk = torch.randn(B, cfg.hidden, generator=gen) if cfg.multi_speaker else None
    t = torch.full((B,), 1095.5)
    prev = lib.cmtts_set_persistent_denoiser(2)
    model.set_precision(dtype)
    try:
        one = model.net(x, t, cond, spk)
        one2 = model.net(x, t, cond, spk)            # back to back: the granule slots are cleared per launch
        lib.cmtts_set_persistent_denoiser(0)
        ref = model.net(x, t, cond, spk)
    finally:
        lib.cmtts_set_persistent_denoiser(prev)
        model.set_precision("fp32")
    host.synchronize()
    assert torch.isfinite(one).all()
    assert torch.equal(one, ref), float((one - ref).abs().max())
    assert torch.equal(one2, ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vocoder_mrf_streams_bitwise(dtype):
    """Small batches: the three ResBlocks of an MRF stage run on three streams, their sum still accumulates in ResBlock
    order -> bit-identical to the in-line order, call after call."""
    host = _host()
    lib = _lib.load()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=5))
    voc.set_precision(dtype)
    mel = (torch.randn(2, 80, 61, generator=torch.Generator().manual_seed(2)) * 1.5 - 4).to(DEV)
    prev = lib.cmtts_set_option(b"branch_streams", 0)
    try:
        ref = voc(mel).clone()
        lib.cmtts_set_option(b"branch_streams", 1)
        for _ in range(4):
            got = voc(mel)
            torch.cuda.synchronize()
            assert torch.equal(got, ref)
    finally:
        lib.cmtts_set_option(b"branch_streams", prev)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T", [(2, 61), (1, 7), (40, 33)])
def test_vocoder_pair16_kernel_bitwise(B, T, dtype):
    """resblock_pair16.hip (16-bit operands, C = 64 / 32 stages, one launch per ResBlock pair) and its conv_xl16_kernel (C = 128 /
    256 stages, one X-resident launch per conv): same conversions, same (chunk, tap, k-group) accumulation order and epilogue
    as conv_mfma16.hip -> bitwise equal to the chunked two-launch 16-bit path."""
    host = _host()
    lib = _lib.load()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=6))
    voc.set_precision(dtype)
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(T)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_pair", 0)
    prev_x = _lib.internal_set(b"voc_xl16", 0)
    prev_r = _lib.internal_set(b"voc_rb16", 2)      # 2 = for every (C, k) of the narrow stages
    try:
        got_rb = voc(mel).clone()                     # C = 64 / 32 stages: a whole ResBlock (three pairs) per launch
        _lib.internal_set(b"voc_rb16", 0)
        ref = voc(mel).clone()                        # every ResBlock conv on the chunked conv_mfma16 kernel
        _lib.internal_set(b"voc_xl16", 1)          # C = 128 / 256 stages on the X-resident conv_xl16 kernel
        got_x = voc(mel).clone()
        _lib.internal_set(b"voc_pair", 2)          # 2 = every (C, k) through the per-tile streamed pair kernel
        got = voc(mel).clone()
        got2 = voc(mel).clone()                       # back to back: stale LDS / staging state must not leak
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"voc_pair", prev)
        _lib.internal_set(b"voc_xl16", prev_x)
        _lib.internal_set(b"voc_rb16", prev_r)
    assert torch.equal(got_rb, ref), float((got_rb - ref).abs().max())
    assert torch.isfinite(got).all()
    assert torch.equal(got_x, ref), float((got_x - ref).abs().max())
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert torch.equal(got2, ref)


@pytest.mark.parametrize("B,T", [(2, 61), (1, 7), (3, 130), (1, 1), (5, 300)])
def test_vocoder_pair16x3_bitwise(B, T):
    """resblock_pair16x3.hip (fp16x3 operands: one X-resident launch per ResBlock pair of the C = 128 / 64 / 32 stages, (hi, lo) LDS
    images of x and xt) keeps conv_mfma16.hip MODE 3's split arithmetic, (chunk, tap, k-group) order with the small terms first and
    its epilogue expressions: the wav must not change by a bit against two chunked launches per pair.  T = 1 / 7: tiles that are all
    halo; 61 / 130: ragged last tiles at every stage (x 8, x 64, x 128, x 256 frames); 5 x 300: thousands of tiles per launch, workgroup counts that
    are not multiples of 8 (the XCD-aware tile order must stay a bijection)."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=9))
    voc.set_precision("fp16x3")
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(70 + T)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_pair3", 0)
    try:
        ref = voc(mel).clone()
        _lib.internal_set(b"voc_pair3", 1)
        got = voc(mel).clone()
        got2 = voc(mel).clone()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"voc_pair3", prev)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert torch.equal(got2, ref)


@pytest.mark.parametrize("B,T", [(2, 61), (3, 130), (1, 1)])
def test_vocoder_fp16x3_upsamplers(B, T):
    """fp16x3 upsamplers (convT_xl16_kernel MODE 3: (hi, lo) images and fragment sets, three fp16 MFMAs per product) against the fp32
    upsamplers of the same precision mode ("ups16" = 0) and against the all-fp32 generator: fp32-class agreement, not the same bits."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=10))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(90 + T)) * 1.5 - 4).to(DEV)
    ref32 = voc(mel).clone()
    voc.set_precision("fp16x3")
    got = voc(mel).clone()
    prev = voc.set_option("ups16", 0)
    try:
        ref = voc(mel).clone()
    finally:
        voc.set_option("ups16", prev)
    again = voc(mel).clone()
    torch.cuda.synchronize()
    scale = float(ref32.abs().max())
    assert torch.isfinite(got).all() and torch.equal(again, got)
    assert not torch.equal(got, ref)
    assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1.0), (float((got - ref).abs().max()), scale)
    assert float((got - ref32).abs().max()) <= 2e-5 * max(scale, 1.0), (float((got - ref32).abs().max()), scale)


@pytest.mark.parametrize("B,T", [(2, 61), (1, 7), (3, 130), (1, 1)])
def test_vocoder_conv_post_v4_bitwise(B, T):
    """conv_post_v4_kernel (three 16-byte loads per channel instead of ten scalars; four channels in flight) keeps the (channel, tap)
    accumulation order of conv_post_kernel: the wav must not change by a bit (first / last tiles with clamped neighbour loads included)."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=11))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(110 + T)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"post_v4", 0)
    try:
        ref = voc(mel).clone()
        _lib.internal_set(b"post_v4", 1)
        got = voc(mel).clone()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"post_v4", prev)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())


@pytest.mark.parametrize("B,T", [(2, 61), (1, 7), (3, 130), (1, 1)])
def test_vocoder_upsampler_kernel_bitwise(B, T):
    """convT_xl_kernel (all stride phases of a HiFi-GAN ConvTranspose1d in one X-resident launch: a two-tap conv with s * C_out
    stacked rows and a phase-interleaving store) against the generic kernel run once per phase: same staging arithmetic (x / 3,
    LeakyReLU), tap order, accumulation order and epilogue -> the wav must not change by a bit; T = 1 and 7 exercise tiles that
    are all halo, 61 / 130 ragged last tiles."""
    host = _host()
    lib = _lib.load()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=8))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(50 + T)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_upsT", 0)
    try:
        ref = voc(mel).clone()
        _lib.internal_set(b"voc_upsT", 1)
        got = voc(mel).clone()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"voc_upsT", prev)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())


@pytest.mark.parametrize("B,T", [(2, 61), (1, 7), (3, 130)])
def test_vocoder_pair_kernel_bitwise(B, T):
    """resblock_pair.hip (C = 64 / 32 stages: conv1 -> LeakyReLU -> conv2 -> + x of a ResBlock pair in one launch, the x
    tile and xt on chip) keeps the generic kernel's (chunk, tap, k) accumulation order and epilogue expressions: the
    wav must not change by a bit against the two-launch path.  T = 61 / 130 mel frames = many 256-column tiles per
    utterance with ragged last tiles; T = 7 = utterances shorter than one tile (every column is halo or padding)."""
    host = _host()
    lib = _lib.load()
    hcfg = HifiGanConfig()
    hsd = synth_hifigan_state_dict(hcfg, seed=6)
    voc = host.Generator(hcfg, DEV).load_state_dict(hsd)
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(T)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_pair", 0)
    prev_x = _lib.internal_set(b"voc_xl", 0)
    prev_b = lib.cmtts_set_option(b"branch_streams", 0)
    try:
        ref = voc(mel).clone()
        _lib.internal_set(b"voc_pair", 1)
        _lib.internal_set(b"voc_xl", 1)        # C = 128 / 256 stages: X-resident single convs (conv_xl_kernel)
        got = voc(mel).clone()
        lib.cmtts_set_option(b"branch_streams", 1)       # the three ResBlock chains of a stage on three streams
        got_s = voc(mel).clone()
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"voc_pair", prev)
        _lib.internal_set(b"voc_xl", prev_x)
        lib.cmtts_set_option(b"branch_streams", prev_b)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert torch.equal(got_s, ref), float((got_s - ref).abs().max())
    if T == 61:
        refo = O.hifigan_generator(hsd, hcfg, _np(mel))
        assert np.abs(_np(got) - refo).max() < 1e-4


@pytest.mark.parametrize("B,T", [(1, 150), (2, 77), (5, 33)])
def test_vocoder_xl_split_bitwise(B, T, voc_form):
    """Round 4: a request or two through the C = 256 stage is a few 64-column tiles; conv_xl then spreads a tile's eight m-tiles over four
    2-wave workgroups (one wave per SIMD, four times the CUs) instead of one 8-wave workgroup.  A wave's accumulation chains and the
    epilogue are unchanged: the wav keeps its bits, and an utterance still equals its row of a chip-filling batch."""
    host = _host()
    hcfg = HifiGanConfig()
    voc = host.Generator(hcfg, DEV).load_state_dict(synth_hifigan_state_dict(hcfg, seed=8))
    mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(T + B)) * 1.5 - 4).to(DEV)
    prev = _lib.internal_set(b"voc_xl_split", 1)
    try:
        got = voc(mel).clone()
        _lib.internal_set(b"voc_xl_split", 0)
        ref = voc(mel).clone()
    finally:
        _lib.internal_set(b"voc_xl_split", prev)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and torch.equal(got, ref), float((got - ref).abs().max())
    big = voc(mel.repeat(24 // B + 1, 1, 1))          # > 80 tiles in the C = 256 stage: the unsplit form
    torch.cuda.synchronize()
    assert same_wav(big[:B], got, voc_form), float((big[:B] - got).abs().max())      # (Winograd form: the batch's C = 128 stage takes it)


def test_hifigan_vs_oracle_other_shape():
    host = _host()
    hcfg = HifiGanConfig()
    hsd = synth_hifigan_state_dict(hcfg, seed=3)
    voc = host.Generator(hcfg, DEV).load_state_dict(hsd)
    rs = np.random.RandomState(9)
    mel = (rs.standard_normal(size=(3, 80, 33)) * 1.5 - 4.0).astype(np.float32)
    wav = voc(torch.from_numpy(mel))
    torch.cuda.synchronize()
    ref = O.hifigan_generator(hsd, hcfg, mel)
    assert np.abs(_np(wav) - ref).max() < 1e-4
    # batch independence at a second size
    one = voc(torch.from_numpy(mel[1:2]))
    assert torch.equal(one[0], wav[1])


def test_errors_are_loud():
    host = _host()
    cfg = get_config("VCTK")
    model = host.CMTotalTTS(cfg, DEV)
    with pytest.raises(RuntimeError):
        model.net(torch.zeros(1, 1, 8, 80), torch.ones(1), torch.zeros(1, 8, 256), torch.zeros(1, 256))
    sd = synth_cmtts_state_dict(cfg, seed=1)
    bad = dict(sd)
    del bad["net.skip_projection.conv.weight"]
    with pytest.raises(RuntimeError, match="missing tensor"):
        host.CMTotalTTS(cfg, DEV).load_state_dict(bad)
    model.load_state_dict(sd)
    with pytest.raises(AssertionError):      # model/cmtts.py:80
        model.duration_pitch_energy_net(None, torch.ones(1, 4, dtype=torch.long), torch.tensor([4]))


def _pack_wino43(w):
    """w [Cout][Cin][k] (torch layout) -> conv_xlq_kernel's weight stream (cmtts_api.hip: to_wino43_iter_fragments; conv_xlq.hip: QTab<k>):
    [Cin/4 k-steps][Cout/64 waves][points][64 lanes][4], element i at lane l = input channel 4 ks + (l >> 4), output row 64 w + 16 i + (l & 15)."""
    from oracle import winograd_ref as W
    cout, cin, k = w.shape
    wz = np.concatenate([w.astype(np.float64), np.zeros((cout, cin, 2))], axis=2)
    pts = []                                                      # every point's [Cout][Cin] transformed weights, in stream order
    for kind, o in W.F43_TAPS[k]:
        if kind == "f43":
            pts += W.f43_weights(wz[:, :, o:o + 3])
        else:
            g = wz[:, :, o]
            pts += [g, 0.5 * g, 0.5 * g, g]
    P = np.stack(pts).astype(np.float32)                          # [npt][Cout][Cin]
    lane = np.arange(64)
    out = np.empty((cin // 4, cout // 64, len(pts), 64, 4), np.float32)
    for ks in range(cin // 4):
        for wv in range(cout // 64):
            for i in range(4):
                out[ks, wv, :, :, i] = P[:, 64 * wv + 16 * i + (lane & 15), 4 * ks + (lane >> 4)]
    return out.reshape(-1)


@pytest.mark.parametrize("Cc,k,dil,T,ld", [(64, 7, 1, 131, 131), (64, 11, 3, 200, 203), (128, 3, 1, 66, 68), (128, 7, 1, 257, 260), (128, 11, 5, 130, 130),
                                            (128, 3, 5, 59, 64), (256, 11, 1, 64, 64), (256, 7, 3, 121, 124), (256, 3, 1, 5, 8)])
def test_conv_xlq_kernel_vs_oracle(Cc, k, dil, T, ld):
    """conv_xlq_kernel (round 5: HiFi-GAN's ResBlock convs as F(4,3) tap groups over output quads) called alone against the oracle — the plain
    conv in float64 (oracle/winograd_ref.py: conv1d_direct) and the restatement of the kernel's own products in fp32 (conv1d_f43_taps at dilation
    1): LeakyReLU on the input, bias, residual and y += (the MRF sum) in the epilogue, lengths that leave a ragged quad and a ragged tile at every
    dilation, row strides that are not a multiple of 16 bytes (the generator itself only ever presents whole quads and aligned rows, so this is
    the only place the scalar epilogue runs at dilation 1)."""
    import ctypes as C
    from oracle import winograd_ref as W
    lib = C.CDLL(_lib.LIB_PATH)

    class XlArgs(C.Structure):
        _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                    ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                    ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long),
                    ("wino_force", C.c_int)]
    lib.cmtts_launch_conv_xlq.restype = C.c_int
    rs = np.random.RandomState(Cc + 10 * k + dil)
    B = 2
    x = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    res = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    y0 = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    w = (rs.standard_normal((Cc, Cc, k)) / np.sqrt(Cc * k)).astype(np.float32)
    bias = rs.standard_normal(Cc).astype(np.float32)
    xd, rd, bd = (torch.from_numpy(v).to(DEV) for v in (x, res, bias))
    wf = torch.from_numpy(_pack_wino43(w)).to(DEV)
    act = np.where(x > 0, x, x * np.float32(0.1))[:, :, :T]
    for accum in (0, 1):
        yd = torch.from_numpy(y0).to(DEV)
        a = XlArgs(xd.data_ptr(), yd.data_ptr(), wf.data_ptr(), bd.data_ptr(), rd.data_ptr(), Cc * ld, B, Cc, T, ld, k, dil, accum, 0.1, 0, 0, 0, 1)
        assert lib.cmtts_launch_conv_xlq(C.byref(a), None) == 0
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        assert np.array_equal(got[:, :, T:], y0[:, :, T:])                  # nothing written beyond T
        for b in range(B):
            ref = W.conv1d_direct(act[b].astype(np.float64), w.astype(np.float64), dil) + bias[:, None] + res[b, :, :T]
            if accum:
                ref = ref + y0[b, :, :T]
            err = np.abs(got[b, :, :T] - ref).max()
            assert err < 2e-5, (accum, b, err)
            if dil == 1 and not accum:      # the restatement of the kernel's own products, fp32: same sums up to the MFMA's accumulation order
                own = W.conv1d_f43_taps(act[b], w) + bias[:, None] + res[b, :, :T]
                assert np.abs(got[b, :, :T] - own).max() < 3e-5      # (measured 1.3e-5 at C = 256: numpy's and the MFMA's K = 256 sums round differently)


@pytest.mark.parametrize("Cc,dil,T,ld", [(128, 1, 66, 68), (128, 3, 200, 203), (128, 5, 59, 64), (128, 1, 257, 260), (128, 5, 512, 512), (64, 1, 131, 131), (64, 3, 5, 8),
                                          (64, 5, 300, 300), (128, 3, 1, 4), (64, 1, 60, 60), (128, 3, 56, 56), (64, 5, 113, 116)])
def test_conv_xlq_pair_vs_oracle(Cc, dil, T, ld):
    """Round 6 (VERDICT r05 #4 ii): conv_xlq_pair3_kernel — a k = 3 ResBlock pair (conv1 at dilation 1 / 3 / 5 -> LeakyReLU -> conv2 -> + x [+ the MRF sum]) with
    both convs in the F(4,3) form in ONE launch, xt in LDS, conv2's halo recomputed — against the oracle's plain pair in float64 (oracle/winograd_ref.py:
    conv1d_direct) and against the two conv_xlq launches it replaces (the same F(4,3) products on quads that start one frame earlier — the fused tile's conv1
    covers [t0 - 1, ..) — so the two agree to fp32 Winograd rounding, not bit for bit): every dilation, ragged quads / tiles (60- and 56-frame tiles), one-frame
    and one-tile inputs, rows that are not 16-byte aligned, with and without accumulation; an utterance alone gets the bits it has in the batch."""
    import ctypes as C
    from oracle import winograd_ref as W
    lib = C.CDLL(_lib.LIB_PATH)

    class XlArgs(C.Structure):
        _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                    ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                    ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long),
                    ("wino_force", C.c_int), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("row_split", C.c_int)]

    class PairArgs(C.Structure):
        _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("w1f", C.c_void_p), ("b1", C.c_void_p), ("w2f", C.c_void_p), ("b2", C.c_void_p),
                    ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int), ("dil", C.c_int),
                    ("accum", C.c_int), ("slope", C.c_float), ("dbg", C.c_void_p)]
    lib.cmtts_launch_conv_xlq.restype = C.c_int
    lib.cmtts_launch_conv_xlq_pair.restype = C.c_int
    rs = np.random.RandomState(Cc + 7 * dil + T)
    B = 2
    x = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    y0 = rs.standard_normal((B, Cc, ld)).astype(np.float32)
    w1 = (rs.standard_normal((Cc, Cc, 3)) / np.sqrt(Cc * 3)).astype(np.float32)
    w2 = (rs.standard_normal((Cc, Cc, 3)) / np.sqrt(Cc * 3)).astype(np.float32)
    b1 = rs.standard_normal(Cc).astype(np.float32)
    b2 = rs.standard_normal(Cc).astype(np.float32)
    xd, b1d, b2d = (torch.from_numpy(v).to(DEV) for v in (x, b1, b2))
    w1f = torch.from_numpy(_pack_wino43(w1)).to(DEV)
    w2f = torch.from_numpy(_pack_wino43(w2)).to(DEV)
    lk = lambda v: np.where(v > 0, v, v * 0.1)
    for accum in (0, 1):
        # the two-launch form: xt = conv1(leaky(x)) + b1 through HBM, y = (conv2(leaky(xt)) + b2) + x [+ y_old]
        xt = torch.full((B, Cc, ld), 7.0, device=DEV)
        yref = torch.from_numpy(y0).to(DEV)
        a1 = XlArgs(xd.data_ptr(), xt.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), None, Cc * ld, B, Cc, T, ld, 3, dil, 0, 0.1, 0, 0, 0, 1, None, None, 0.0, 0)
        assert lib.cmtts_launch_conv_xlq(C.byref(a1), None) == 0
        a2 = XlArgs(xt.data_ptr(), yref.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), xd.data_ptr(), Cc * ld, B, Cc, T, ld, 3, 1, accum, 0.1, 0, 0, 0, 1, None, None, 0.0, 0)
        assert lib.cmtts_launch_conv_xlq(C.byref(a2), None) == 0
        yd = torch.from_numpy(y0).to(DEV)
        pa = PairArgs(xd.data_ptr(), yd.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), Cc * ld, B, Cc, T, ld, 3, dil, accum, 0.1, None)
        assert lib.cmtts_launch_conv_xlq_pair(C.byref(pa), None) == 0
        torch.cuda.synchronize()
        got, two = yd.cpu().numpy(), yref.cpu().numpy()
        assert np.array_equal(got[:, :, T:], y0[:, :, T:])                  # nothing written beyond T
        assert np.abs(got - two).max() < 1e-5, (accum, float(np.abs(got - two).max()))      # (measured <= 4.3e-6 on outputs of a few units)
        if not accum:
            y1 = torch.from_numpy(y0[1:]).to(DEV)
            pa1 = PairArgs(xd[1:].data_ptr(), y1.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), Cc * ld, 1, Cc, T, ld, 3, dil, 0, 0.1, None)
            assert lib.cmtts_launch_conv_xlq_pair(C.byref(pa1), None) == 0
            torch.cuda.synchronize()
            assert np.array_equal(y1.cpu().numpy()[0], got[1])
        for b in range(B):
            xt64 = W.conv1d_direct(lk(x[b, :, :T].astype(np.float64)), w1.astype(np.float64), dil) + b1[:, None]
            ref = W.conv1d_direct(lk(xt64), w2.astype(np.float64), 1) + b2[:, None] + x[b, :, :T]
            if accum:
                ref = ref + y0[b, :, :T]
            err = np.abs(got[b, :, :T] - ref).max()
            assert err < 3e-5, (accum, b, err)
    # the shapes the kernel does not take are refused, not mangled
    pa = PairArgs(xd.data_ptr(), yd.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), Cc * ld, B, Cc, T, ld, 7, dil, 0, 0.1, None)
    assert lib.cmtts_launch_conv_xlq_pair(C.byref(pa), None) == -2
    pa = PairArgs(xd.data_ptr(), yd.data_ptr(), w1f.data_ptr(), b1d.data_ptr(), w2f.data_ptr(), b2d.data_ptr(), Cc * ld, B, Cc, T, ld, 3, 2, 0, 0.1, None)
    assert lib.cmtts_launch_conv_xlq_pair(C.byref(pa), None) == -2


@pytest.mark.parametrize("cin,T,ld,B,ln", [(128, 131, 132, 2, False), (256, 200, 203, 2, True), (256, 64, 64, 3, False), (256, 5, 8, 1, True), (128, 513, 516, 2, False),
                                           (256, 512, 512, 4, True)])
def test_conv_k5q_kernel_vs_oracle(cin, T, ld, B, ln):
    """Round 6: conv_k5q_kernel (the frame-level pitch predictor's Conv1d(cin -> 256, k = 5) + bias + ReLU as two F(4,3) tap groups over frame quads,
    optionally with the previous block's LayerNorm applied to the staged tile) called alone against the oracle: the plain conv in float64 on the
    float64 LayerNorm (oracle/winograd_ref.py: conv1d_direct) and the restatement of the kernel's own products in fp32 (conv1d_f43_taps, k = 5);
    ragged quads / tiles, rows that are not 16-byte aligned (scalar epilogue), and every row split (1 / 2 / 4 workgroups per tile) bit for bit."""
    import ctypes as C
    from oracle import winograd_ref as W
    lib = C.CDLL(_lib.LIB_PATH)

    class XlArgs(C.Structure):
        _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                    ("bstride", C.c_long), ("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("ld", C.c_int), ("k", C.c_int),
                    ("dil", C.c_int), ("accum", C.c_int), ("slope", C.c_float), ("relu", C.c_int), ("cin", C.c_int), ("xbstride", C.c_long),
                    ("wino_force", C.c_int), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("row_split", C.c_int)]
    lib.cmtts_launch_conv_k5q.restype = C.c_int
    rs = np.random.RandomState(cin + T)
    x = (rs.standard_normal((B, cin, ld)) * 1.5 + 0.3).astype(np.float32)
    w = (rs.standard_normal((256, cin, 5)) / np.sqrt(cin * 5)).astype(np.float32)
    bias = rs.standard_normal(256).astype(np.float32)
    g = (1.0 + 0.1 * rs.standard_normal(256)).astype(np.float32)
    be = (0.1 * rs.standard_normal(256)).astype(np.float32)
    xd, bd, gd, bed = (torch.from_numpy(v).to(DEV) for v in (x, bias, g, be))
    wf = torch.from_numpy(_pack_wino43(w)).to(DEV)
    y0 = rs.standard_normal((B, 256, ld)).astype(np.float32)
    outs = []
    for split in (0, 1, 2, 4):
        yd = torch.from_numpy(y0).to(DEV)
        a = XlArgs(xd.data_ptr(), yd.data_ptr(), wf.data_ptr(), bd.data_ptr(), None, 256 * ld, B, 256, T, ld, 5, 1, 0, 1.0, 1, 0 if cin == 256 else cin, cin * ld,
                   1, gd.data_ptr() if ln else None, bed.data_ptr() if ln else None, 1e-12, split)
        assert lib.cmtts_launch_conv_k5q(C.byref(a), None) == 0
        torch.cuda.synchronize()
        outs.append(yd.cpu().numpy())
    got = outs[0]
    for o in outs[1:]:
        assert np.array_equal(o, got)                                       # the row split never changes a bit
    assert np.array_equal(got[:, :, T:], y0[:, :, T:])                      # nothing written beyond T
    for b in range(B):
        xin64 = x[b, :, :T].astype(np.float64)
        xin32 = x[b, :, :T]
        if ln:
            mu, var = xin64.mean(0), xin64.var(0)
            xin64 = (xin64 - mu) / np.sqrt(var + 1e-12) * g[:, None] + be[:, None]
            xin32 = xin64.astype(np.float32)
        ref = np.maximum(W.conv1d_direct(xin64, w.astype(np.float64), 1) + bias[:, None], 0.0)
        err = np.abs(got[b, :, :T] - ref).max()
        assert err < 2e-5, (b, err)
        own = np.maximum(W.conv1d_f43_taps(xin32, w) + bias[:, None], 0.0)
        assert np.abs(got[b, :, :T] - own).max() < 3e-5


@pytest.mark.parametrize("variant,B,L", [("LJSpeech", 32, 85), ("VCTK", 1, 25), ("LibriTTS", 3, 40)])
def test_pitch_predictor_winograd(variant, B, L):
    """Round 6: the frame-level pitch predictor with its k = 5 convs as F(4,3) tap groups (conv_k5q.hip, the default at every batch size) against
    the direct form (pred_wino = 0: conv_xl / conv_xres / generic kernels): cwt output within fp32 Winograd rounding, durations and mel2ph
    untouched (they are upstream), pitch buckets equal on these batches; an utterance alone gets the bits it has inside the batch."""
    host = _host()
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=13, dur_frames=6.0, dur_spread=0.0))
    rs = np.random.RandomState(500 + L)
    texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64))
    lens = torch.full((B,), L, dtype=torch.int64)
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    T = 6 * L
    prev = _lib.internal_set(b"pred_wino", 0)
    try:
        ref = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=T)
        ref = {"cwt": ref["p_predictions"]["cwt"].clone(), "p_idx": ref["p_predictions"]["p_idx"].clone(), "mel2ph": ref["mel2ph"].clone(),
               "cond_ct": ref["cond_ct"].clone(), "f0": ref["p_predictions"]["f0_denorm"].clone()}
        assert _lib.internal_set(b"pred_wino", 1) == 0
        out = model.duration_pitch_energy_net(None, texts, lens, spker_embeds=spk, max_mel_len=T)
        one = model.duration_pitch_energy_net(None, texts[B - 1:], lens[B - 1:], spker_embeds=None if spk is None else spk[B - 1:], max_mel_len=T)
        torch.cuda.synchronize()
    finally:
        _lib.internal_set(b"pred_wino", prev)
    d = float((out["p_predictions"]["cwt"] - ref["cwt"]).abs().max())
    scale = float(ref["cwt"].abs().max())
    flips = int((out["p_predictions"]["p_idx"] != ref["p_idx"]).sum())
    report(f"PRED_WINO {variant} B={B} L={L}: max|d cwt| {d:.2e} on |cwt| <= {scale:.2f}; pitch-bucket flips vs the direct form {flips} of {ref['p_idx'].numel()}")
    assert 0 < d < 2e-5 * max(1.0, scale)
    assert torch.equal(out["mel2ph"], ref["mel2ph"])
    # unsearched inputs (no margins): a frame whose pre-rounding bucket value lies on a rounding boundary may go to the neighbouring bucket — counted,
    # required to sit on a boundary and to be off by one; everywhere else the conditioning is the direct form's, bit for bit
    from conftest import FLIP_MARGIN
    same = _np(out["p_predictions"]["p_idx"]) == _np(ref["p_idx"])
    on_boundary = ~pitch_margin_mask(_np(ref["f0"]), FLIP_MARGIN)
    assert flips <= 2 and not (~same & ~on_boundary).any()
    assert (np.abs(_np(out["p_predictions"]["p_idx"]) - _np(ref["p_idx"])) <= 1).all()
    same_t = torch.from_numpy(same).to(out["cond_ct"].device)[:, None, :]
    assert torch.equal(torch.where(same_t, out["cond_ct"], torch.zeros_like(out["cond_ct"])), torch.where(same_t, ref["cond_ct"], torch.zeros_like(ref["cond_ct"])))
    assert torch.equal(one["p_predictions"]["cwt"][0], out["p_predictions"]["cwt"][B - 1])      # one form at every batch size
    assert torch.equal(one["cond_ct"][0], out["cond_ct"][B - 1])


@pytest.mark.parametrize("layers", [1, 2, 3])
def test_persistent_stack_few_layers(layers):
    """The persistent stacks with one, two and three residual layers — the first layer is then (also) the last one: no publish phase, only the
    skip half of the output projection (round 5), the in-kernel tail straight behind it.  Forced onto a small batch: the direct form bit for bit
    the per-layer kernels (fp32 and bf16), both Winograd forms within WINO_TOL, one evaluation and a T = 2 sample."""
    import dataclasses
    host = _host()
    lib = _lib.load()
    cfg = dataclasses.replace(get_config("VCTK"), res_layers=layers)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=layers))
    B, T = 3, 130
    g = torch.Generator().manual_seed(layers)
    x = torch.randn(B, 1, T, 80, generator=g).to(DEV); cond = torch.randn(B, T, 256, generator=g).to(DEV); spk = torch.randn(B, 256, generator=g).to(DEV)
    t = torch.full((B,), 1095.5, device=DEV)
    noise = torch.randn(3, B, 1, T, 80, generator=g).to(DEV)
    cond_ct = cond.transpose(1, 2).contiguous()
    outs, mels = {}, {}
    prev = lib.cmtts_set_persistent_denoiser(0)
    prev_w = _lib.internal_set("persist_wino", 0)
    try:
        for prec in ("fp32", "bf16"):
            model.set_precision(prec)
            lib.cmtts_set_persistent_denoiser(0)
            outs[("layers", prec)] = model.net(x, t, cond, spk).clone()
            mels[("layers", prec)] = host.sample_with_cond(model, cond_ct, spk, 2, noise).clone()
            lib.cmtts_set_persistent_denoiser(2)
            for wn in ((0, 1, 3) if prec == "fp32" else (0,)):
                _lib.internal_set("persist_wino", wn)
                outs[(wn, prec)] = model.net(x, t, cond, spk).clone()
                mels[(wn, prec)] = host.sample_with_cond(model, cond_ct, spk, 2, noise).clone()
            _lib.internal_set("persist_wino", 0)
    finally:
        model.set_precision("fp32")
        _lib.internal_set("persist_wino", prev_w)
        lib.cmtts_set_persistent_denoiser(prev)
    host.synchronize()
    for prec in ("fp32", "bf16"):
        assert torch.equal(outs[(0, prec)], outs[("layers", prec)]) and torch.equal(mels[(0, prec)], mels[("layers", prec)]), prec
    for wn in (1, 3):
        d, dm = float((outs[(wn, "fp32")] - outs[(0, "fp32")]).abs().max()), float((mels[(wn, "fp32")] - mels[(0, "fp32")]).abs().max())
        assert torch.isfinite(outs[(wn, "fp32")]).all() and 0 < d <= WINO_TOL and dm <= WINO_TOL, (wn, d, dm)
