"""Pin the numpy oracle (oracle/cmtts_oracle.py) against golden vectors captured from the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import cmtts_amd
from cmtts_amd.config import get_config, HifiGanConfig
from cmtts_amd.weights import synth_cmtts_state_dict, synth_hifigan_state_dict
from oracle import cmtts_oracle as O
from conftest import golden_noise, pitch_margin_mask

VARIANTS = ["LJSpeech", "VCTK", "LibriTTS"]


def _setup(golden, variant):
    g = golden("cmtts_" + variant)
    cfg = get_config(variant)
    sd = synth_cmtts_state_dict(cfg, seed=int(g["seed"]), dur_frames=4.0, dur_spread=0.03)
    return g, cfg, sd


@pytest.mark.parametrize("variant", VARIANTS)
def test_duration_pitch_speaker_net(golden, variant):
    g, cfg, sd = _setup(golden, variant)
    st = O.duration_pitch_speaker_net(sd, cfg, g["texts"], g["src_lens"], g.get("spker_embeds"))
    np.testing.assert_allclose(st["enc_out"], g["enc_out"], atol=2e-5)
    np.testing.assert_allclose(st["log_d"], g["log_d"], atol=2e-5)
    np.testing.assert_allclose(st["e_pred"], g["e_pred"], atol=5e-5)
    # integer stages: bit-exact
    np.testing.assert_array_equal(st["d_rounded"], g["d_rounded"])
    np.testing.assert_array_equal(st["mel_len"], g["mel_len"])
    np.testing.assert_array_equal(st["mel2ph"], g["mel2ph"])
    L = g["texts"].shape[1]
    valid = np.arange(L)[None, :] < g["src_lens"][:, None]
    np.testing.assert_array_equal(st["e_idx"][valid], g["e_idx"][valid])
    np.testing.assert_array_equal(st["mel_mask"], g["mel_mask"])
    if cfg.multi_speaker:
        np.testing.assert_allclose(st["speaker_emb"], g["speaker_emb"], atol=1e-5)
    np.testing.assert_allclose(st["cwt_out"], g["cwt_out"], atol=1e-4)
    np.testing.assert_allclose(st["f0_mean"], g["f0_mean"], atol=1e-5)
    np.testing.assert_allclose(st["f0_std"], g["f0_std"], atol=1e-5)
    np.testing.assert_allclose(st["f0_denorm"], g["f0_denorm"], rtol=2e-4, atol=1e-3)
    ok = pitch_margin_mask(g["f0_denorm"])
    assert ok.mean() > 0.97
    np.testing.assert_array_equal(st["p_idx"][ok], g["p_idx"][ok])
    same = st["p_idx"] == g["p_idx"]
    np.testing.assert_allclose(st["cond"][same], g["cond"][same], atol=2e-5)


@pytest.mark.parametrize("variant", VARIANTS)
def test_denoiser_forward(golden, variant):
    g, cfg, sd = _setup(golden, variant)
    out = O.denoiser_forward(sd, cfg, g["den_x"], g["den_t"], g["cond"], g.get("speaker_emb"))
    assert out.shape == g["den_out"].shape
    np.testing.assert_allclose(out, g["den_out"], atol=2e-4)      # fp32 roundoff over 20 layers, O(1) values


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_steps", [1, 2, 4])
def test_sampler(golden, variant, n_steps):
    """karras_sample_tts with the encoder hoisted out of the loop == the reference's in-loop call."""
    g, cfg, sd = _setup(golden, variant)
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts(sd, cfg, g["cond"], g.get("speaker_emb"), n_steps, noise)
    np.testing.assert_allclose(mel, g[f"mel_T{n_steps}"], atol=2e-4)   # north-star bound is 1e-3


def test_multistep_schedule_constants():
    cfg = get_config("LJSpeech")
    sig, std = O.multistep_schedule(4, cfg)
    assert all(abs(s - 80.0) < 1e-9 for s in sig)
    assert abs(std[0] - np.sqrt(80.0 ** 2 - 0.002 ** 2) * 0.85) < 1e-9 and 0.0 <= std[-1] < 1e-9   # (tmin^(1/rho))^rho clips a hair above sigma_min
    c_skip, c_out, c_in = O.boundary_scalings(80.0, cfg)
    assert abs(c_in - 0.0124998) < 1e-6 and abs(c_out - 0.499978) < 1e-5 and abs(c_skip - 3.906e-5) < 1e-7


def test_small_semantics():
    bins = np.linspace(-1.5, 8.0, 255).astype(np.float32)
    assert O.bucketize(np.float32([-9, -1.5, 8.0, 9]), bins).tolist() == [0, 0, 254, 255]
    idx, _ = O.f0_to_coarse(np.float32([0, 50, 100, 440, 1100, 5000]))
    assert idx.tolist() == [1, 1, 20, 122, 255, 255]
    assert O.durations_from_log(np.log(np.float32([1.5, 2.5, 3.5, 0.2]))).tolist() == [0, 2, 2, 0]
    assert O.wav_to_int16(np.float32([1.0, -1.0, 0.99999, -0.00002])).tolist() == [-32768, -32768, 32767, 0]


def test_hifigan(golden):
    g = golden("hifigan")
    hcfg = HifiGanConfig()
    hsd = synth_hifigan_state_dict(hcfg, seed=int(g["seed"]))
    pcm, wav = O.vocoder_infer(hsd, hcfg, g["mel"], g["mel_lens"], get_config("LJSpeech"))
    np.testing.assert_allclose(wav, g["wav"][:, 0], atol=2e-5)
    for i, name in enumerate(["pcm0", "pcm1"]):
        assert pcm[i].shape == g[name].shape
        assert np.abs(pcm[i].astype(np.int32) - g[name].astype(np.int32)).max() <= 1


def test_torch_backend_agrees_with_numpy_backend(golden):
    """The optional torch-CPU primitives (used only for the timed cpu_baseline) compute the same graph."""
    g, cfg, sd = _setup(golden, "VCTK")
    ref = O.denoiser_forward(sd, cfg, g["den_x"], g["den_t"], g["cond"], g.get("speaker_emb"))
    O.set_backend("torch")
    try:
        out = O.denoiser_forward(sd, cfg, g["den_x"], g["den_t"], g["cond"], g.get("speaker_emb"))
        st = O.duration_pitch_speaker_net(sd, cfg, g["texts"], g["src_lens"], g.get("spker_embeds"))
    finally:
        O.set_backend("numpy")
    np.testing.assert_allclose(out, ref, atol=2e-4)
    np.testing.assert_allclose(out, g["den_out"], atol=2e-4)
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts_torch(sd, cfg, g["cond"], g.get("speaker_emb"), 2, noise)
    np.testing.assert_allclose(mel, g["mel_T2"], atol=2e-4)
    np.testing.assert_array_equal(st["mel_len"], g["mel_len"])
    np.testing.assert_array_equal(st["mel2ph"], g["mel2ph"])


@pytest.mark.parametrize("sampler", ["euler", "heun", "dpm", "ancestral"])
def test_ode_samplers(golden, sampler):
    """the reference's other sampler loops (sample_euler/heun/dpm/euler_ancestral)
    around the same denoiser, against karras_sample_tts of the reference on the LJSpeech golden model."""
    g, cfg, sd = _setup(golden, "LJSpeech")
    gs = golden("samplers_LJSpeech")
    assert int(gs["seed"]) == int(g["seed"])
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts_ode(sd, cfg, g["cond"], None, sampler, int(gs["steps_" + sampler]), noise)
    ref = gs["mel_" + sampler]
    assert mel.shape == ref.shape
    # the ODE steps divide by sigma down to 0.002-scale values: fp32 roundoff of the denoiser is amplified,
    # so the bound is relative to the output scale (|mel| up to ~10)
    np.testing.assert_allclose(mel, ref, atol=1e-3, rtol=1e-4)


def test_ode_sampler_edm_scalings(golden):
    """KarrasDenoiser.denoise with distillation=False (karras_diffusion.py:81-85,395-398: get_scalings instead of
    the boundary-condition scalings), heun, against the reference run that way."""
    g, cfg, sd = _setup(golden, "LJSpeech")
    gs = golden("samplers_LJSpeech")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts_ode(sd, cfg, g["cond"], None, "heun", int(gs["steps_heun"]), noise, distillation=False)
    # without the boundary shift c_out(sigma_min) != 0, so the last 1/sigma division amplifies the denoiser's fp32
    # roundoff a little more than in test_ode_samplers: 6 of 25920 elements sit at 1.0-1.4e-3
    np.testing.assert_allclose(mel, gs["mel_heun_edm"], atol=3e-3, rtol=1e-4)
    assert np.abs(gs["mel_heun_edm"] - gs["mel_heun"]).max() > 1e-2     # the flag changes the result


def test_multistep_general_ts(golden):
    """stochastic_iterative_sampler (karras_diffusion.py:830-854) on a schedule with interior points,
    steps=4, ts=(0,1,3), against the reference."""
    g, cfg, sd = _setup(golden, "LJSpeech")
    gs = golden("samplers_LJSpeech")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts(sd, cfg, g["cond"], None, None, noise, ts=(0, 1, 3), steps=4)
    assert int(gs["draws_multistep_ts013"]) == 3
    np.testing.assert_allclose(mel, gs["mel_multistep_ts013"], atol=1e-3)


def test_fastspeech_decoder(golden):
    """FastspeechDecoder (model/modules.py:154-165) restated in the oracle against the reference module's output:
    mask derived from zero rows, and an explicit mask over non-zero padding."""
    from cmtts_amd.weights import synth_decoder_state_dict
    g = golden("decoder_LJSpeech")
    cfg = get_config("LJSpeech")
    sd = synth_decoder_state_dict(cfg, seed=int(g["seed"]))
    x, lens = g["x"], g["lens"]
    pad = np.arange(x.shape[1])[None, :] >= lens[:, None]
    xz = x.copy()
    xz[pad] = 0
    np.testing.assert_allclose(O.fastspeech_decoder(sd, cfg, xz), g["y_auto"], atol=2e-5)
    np.testing.assert_allclose(O.fastspeech_decoder(sd, cfg, x, pad), g["y_mask"], atol=2e-5)
    assert np.abs(g["y_auto"][pad]).max() == 0


def test_sigmas_karras():
    s = O.get_sigmas_karras(5, 0.002, 80.0, 7.0)
    assert s.dtype == np.float32 and s.shape == (6,) and s[-1] == 0
    np.testing.assert_allclose(s[0], 80.0, rtol=1e-6)
    np.testing.assert_allclose(s[4], 0.002, rtol=1e-5)
    assert np.all(np.diff(s) < 0)


def _check_variance(st, gc, tag, g, cfg):
    np.testing.assert_allclose(st["log_d"], gc[tag + "_log_d"], atol=2e-5)
    np.testing.assert_array_equal(st["d_rounded"], gc[tag + "_d_rounded"])
    np.testing.assert_array_equal(st["mel_len"], gc[tag + "_mel_len"])
    np.testing.assert_allclose(st["e_pred"], gc[tag + "_e_pred"], atol=5e-5)
    np.testing.assert_allclose(st["cwt_out"], gc[tag + "_cwt_out"], atol=2e-4)
    np.testing.assert_allclose(st["f0_denorm"], gc[tag + "_f0_denorm"], rtol=2e-4, atol=1e-3)
    ok = pitch_margin_mask(gc[tag + "_f0_denorm"])
    assert ok.mean() > 0.95
    np.testing.assert_array_equal(st["p_idx"][ok], gc[tag + "_p_idx"][ok])
    same = st["p_idx"] == gc[tag + "_p_idx"]
    np.testing.assert_allclose(st["cond"][same], gc[tag + "_cond"][same], atol=3e-5)


def test_variance_controls(golden):
    """p/e/d controls (model/modules.py:270,326,369) against the reference on the VCTK golden model."""
    g, cfg, sd = _setup(golden, "VCTK")
    gc = golden("controls_VCTK")
    st = O.duration_pitch_speaker_net(sd, cfg, g["texts"], g["src_lens"], g["spker_embeds"],
                                      p_control=float(gc["p_control"]), e_control=float(gc["e_control"]),
                                      d_control=float(gc["d_control"]))
    _check_variance(st, gc, "ctl", g, cfg)


def test_variance_teacher_forced(golden):
    """Duration / energy / pitch targets (model/modules.py:318-328,365-367,379-390)."""
    g, cfg, sd = _setup(golden, "VCTK")
    gc = golden("controls_VCTK")
    pt = dict(cwt_spec=gc["tf_cwt_spec"], f0_mean=gc["tf_f0_mean"], f0_std=gc["tf_f0_std"], uv=gc["tf_uv"])
    st = O.duration_pitch_speaker_net(sd, cfg, g["texts"], g["src_lens"], g["spker_embeds"],
                                      max_mel_len=gc["tf_cwt_spec"].shape[1], d_target=gc["tf_d_target"],
                                      e_target=gc["tf_e_target"], pitch_target=pt)
    _check_variance(st, gc, "tf", g, cfg)


def test_precision_modes_of_the_oracle(golden):
    """The float64 mode (the yardstick of the reduced-precision GPU tests) and the 16-bit-operand modes of the oracle:
    f64 agrees with the reference's fp32 golden to fp32 roundoff and the oracle's own fp32 run is no farther from f64
    than the reference's; bf16 / fp16 operand rounding perturbs the mel by about 10 / 1 unit roundoffs of the type."""
    g, cfg, sd = _setup(golden, "VCTK")
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    with O.precision("f64"):
        m64 = O.karras_sample_tts(sd, cfg, g["cond"], g["speaker_emb"], 4, noise)
        assert m64.dtype == np.float64
        with O.operands16("bf16"):
            mb = O.karras_sample_tts(sd, cfg, g["cond"], g["speaker_emb"], 4, noise)
        with O.operands16("fp16"):
            mh = O.karras_sample_tts(sd, cfg, g["cond"], g["speaker_emb"], 4, noise)
    assert O.F32 is np.float32 and O._OPERAND16 is None
    m32 = O.karras_sample_tts(sd, cfg, g["cond"], g["speaker_emb"], 4, noise)
    e_ref, e_own = np.abs(g["mel_T4"] - m64).max(), np.abs(m32 - m64).max()
    e_b, e_h = np.abs(mb - m64).max(), np.abs(mh - m64).max()
    print(f"vs f64: reference fp32 {e_ref:.2e}, oracle fp32 {e_own:.2e}, bf16 operands {e_b:.2e}, fp16 operands {e_h:.2e}")
    assert e_ref < 2e-4 and e_own < 2e-4
    ub, uh = 2.0 ** -8, 2.0 ** -11
    assert 1 * ub < e_b < 20 * ub and 1 * uh < e_h < 20 * uh
    # quant16 is round-to-nearest-even from the fp32 value
    q = O.quant16(np.float32([1.0, 1.00390625, 1.01171875, 65504.0]), "bf16")      # two ties (-> even), one overflow of the mantissa
    assert q.tolist() == [1.0, 1.0, 1.015625, 65536.0]
    assert O.quant16(np.float32([1.00048828125, 1.00146484375]), "fp16").tolist() == [1.0, 1.001953125]


def test_speaker_table_variant(golden):
    """speaker_embedder "none" (model/cmtts.py:26-38,77-78): the speaker vector is a row of the nn.Embedding table."""
    g = golden("cmtts_VCTK_table")
    cfg = get_config("VCTK_table")
    sd = synth_cmtts_state_dict(cfg, seed=int(g["seed"]), dur_frames=4.0, dur_spread=0.03)
    assert sd["duration_pitch_energy_net.speaker_emb.weight"].shape == (cfg.n_speaker, cfg.hidden)
    st = O.duration_pitch_speaker_net(sd, cfg, g["texts"], g["src_lens"], speakers=g["speakers"])
    np.testing.assert_array_equal(st["speaker_emb"], g["speaker_emb"])          # a gather: bit-exact
    np.testing.assert_allclose(st["log_d"], g["log_d"], atol=2e-5)
    np.testing.assert_array_equal(st["d_rounded"], g["d_rounded"])
    np.testing.assert_array_equal(st["mel_len"], g["mel_len"])
    np.testing.assert_array_equal(st["mel2ph"], g["mel2ph"])
    same = st["p_idx"] == g["p_idx"]
    assert same.mean() > 0.97
    np.testing.assert_allclose(st["cond"][same], g["cond"][same], atol=2e-5)
    B, T, _ = g["cond"].shape
    noise = golden_noise(int(g["seed"]), (B, 1, T, cfg.n_mels), 5)
    mel = O.karras_sample_tts(sd, cfg, g["cond"], g["speaker_emb"], 2, noise)
    np.testing.assert_allclose(mel, g["mel_T2"], atol=2e-4)
