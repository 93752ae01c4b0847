"""CPU-side checks of the Python host layer that need no GPU: the nn.Module containers behave like modules (the
reference's synthesize.py calls model.to(device) / eval() / state_dict() right after load_state_dict, synthesize.py:79-86)."""
import torch

import cmtts_amd
from cmtts_amd.config import get_config


def test_module_tree_has_no_cycle():
    from cmtts_amd import host
    model = host.CMTotalTTS(get_config("VCTK"), device="cpu")       # cmtts_create only: no GPU work
    assert model.to("cpu") is model
    assert isinstance(model.state_dict(), dict)
    assert "FastspeechDecoder" in repr(model) and "CMDenoiserTTS" in repr(model)
    assert model.eval() is model and model.train(False) is model
    model.apply(lambda m: None)
    kids = dict(model.named_children())
    assert set(kids) == {"duration_pitch_energy_net", "net", "decoder"}
    for k in kids.values():
        assert list(k.children()) == []                             # the owner is not a registered child of its parts


def test_karras_denoiser_defaults_match_reference():
    """karras_diffusion.py:36-45: distillation defaults to False; synthesize.py builds it with True."""
    from cmtts_amd import host
    d = host.KarrasDenoiser()
    assert d.distillation is False and d.sigma_data == 0.5 and d.sigma_max == 80.0 and d.sigma_min == 0.002 and d.rho == 7.0
    s = torch.tensor([80.0])
    c_skip, c_out, c_in = d.get_scalings_for_boundary_condition(s)
    assert abs(float(c_in) - 0.0124998) < 1e-6 and abs(float(c_out) - 0.499978) < 1e-5


def _reference_configs(table=False):
    """The three YAML dicts of config/VCTK (restated: only the keys the inference path reads)."""
    pre = {"dataset": "VCTK", "path": {"preprocessed_path": "."},
           "preprocessing": {"speaker_embedder": "none" if table else "DeepSpeaker",
                             "audio": {"sampling_rate": 22050, "max_wav_value": 32768.0},
                             "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
                             "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": 8000},
                             "pitch": {"pitch_type": "cwt", "pitch_norm": "log", "pitch_norm_eps": 1e-9, "use_uv": True, "cwt_scales": -1},
                             "energy": {"feature": "phoneme_level", "normalization": True}}}
    mod = {"external_speaker_dim": 512, "multi_speaker": True,
           "transformer": {"encoder_layer": 4, "encoder_head": 2, "encoder_hidden": 256, "ffn_kernel_size": 9},
           "denoiser": {"residual_layers": 20, "residual_channels": 256},
           "variance_predictor": {"filter_size": 256, "predictor_layers": 2, "predictor_kernel": 5, "cwt_hidden_size": 128,
                                  "cwt_std_scale": 0.8, "dur_predictor_layers": 2, "dur_predictor_kernel": 3},
           "variance_embedding": {"pitch_n_bins": 300, "energy_n_bins": 256},
           "vocoder": {"model": "HiFi-GAN", "speaker": "universal"}}
    tr = {"cm": {"training_mode": "consistency_training", "sigma_min": 0.002, "sigma_max": 80.0, "loss_norm": "l1",
                 "weight_schedule": "uniform"}}
    return pre, mod, tr


def test_synthesizer_with_the_reference_constructor(tmp_path):
    """CMTotalTTSSynthesize(model_path, model_step_num, args, preprocess_config, model_config, train_config) as in
    synthesize.py:35-86: the checkpoint <model_path>/CMDenoiserTTS/model{step:06d}.pt is read with torch.load and loaded by
    key; configuration comes from the three YAML dicts.  No GPU here: the weights are taken (every key accepted), the
    kernels refuse to run."""
    import argparse
    import pytest
    from cmtts_amd import host
    from cmtts_amd.config import get_config, config_from_reference
    from cmtts_amd.weights import synth_cmtts_state_dict
    pre, mod, tr = _reference_configs()
    assert config_from_reference(pre, mod, tr) == get_config("VCTK")          # the YAML route gives the built-in variant
    pre_t, _, _ = _reference_configs(table=True)
    assert config_from_reference(pre_t, mod, tr, n_speaker=108) == get_config("VCTK_table").__class__(
        **{**get_config("VCTK_table").to_dict(), "name": "VCTK"})
    sd = synth_cmtts_state_dict(get_config("VCTK"), seed=1)
    ckpt = tmp_path / "CMDenoiserTTS"
    ckpt.mkdir()
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ckpt / "model000300.pt")
    syn = host.CMTotalTTSSynthesize(str(tmp_path), 300, argparse.Namespace(T=4), pre, mod, tr, p_control=1.1, device="cpu")
    assert syn.CMDenoiserTTS_path.endswith("CMDenoiserTTS/model000300.pt")
    assert syn.model.config == get_config("VCTK") and syn.diffusion.distillation is True and syn.diffusion.loss_norm == "l1"
    assert (syn.p_control, syn.e_control, syn.d_control) == (1.1, 1.0, 1.0)
    dpe, den = syn.model.get_segmentation_model()
    assert dpe is syn.duration_pitch_energy_net and callable(den)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            syn.synthesize((["a"], ["x"], None, torch.ones(1, 4, dtype=torch.long), torch.tensor([4]), 4, torch.zeros(1, 512)))
    tr_bad = {"cm": dict(tr["cm"], training_mode="something")}
    with pytest.raises(ValueError, match="unknown training mode"):
        host.CMTotalTTSSynthesize(str(tmp_path), 300, argparse.Namespace(T=1), pre, mod, tr_bad, device="cpu")
    with pytest.raises(FileNotFoundError):
        host.CMTotalTTSSynthesize(str(tmp_path), 301, argparse.Namespace(T=1), pre, mod, tr, device="cpu")


def test_get_vocoder_reads_the_reference_layout(tmp_path):
    """utils/model.py:155-184: hifigan/config.json + hifigan/generator_<speaker>.pth.tar with ckpt["generator"] holding
    weight-norm pairs.  Only the file handling and key folding run here (no GPU)."""
    import json
    import pytest
    from cmtts_amd import host
    from cmtts_amd.config import HifiGanConfig
    from cmtts_amd.weights import synth_hifigan_state_dict
    (tmp_path / "hifigan").mkdir()
    h = HifiGanConfig()
    json.dump({"num_mels": 80, "upsample_rates": list(h.upsample_rates), "upsample_kernel_sizes": list(h.upsample_kernel_sizes),
               "upsample_initial_channel": 512, "resblock_kernel_sizes": list(h.resblock_kernel_sizes),
               "resblock_dilation_sizes": [list(d) for d in h.resblock_dilation_sizes]}, open(tmp_path / "hifigan" / "config.json", "w"))
    sd = synth_hifigan_state_dict(h, seed=2)
    torch.save({"generator": {k: torch.from_numpy(v) for k, v in sd.items()}}, tmp_path / "hifigan" / "generator_universal.pth.tar")
    cfg = {"vocoder": {"model": "HiFi-GAN", "speaker": "universal"}}
    voc = host.get_vocoder(cfg, "cuda:0" if torch.cuda.is_available() else "cpu", root=str(tmp_path))
    if torch.cuda.is_available():
        assert voc._ready
    else:
        assert not voc._ready                      # weights read, folded and handed to the library; no device to upload to
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            voc(torch.zeros(1, 80, 4))
    with pytest.raises(NotImplementedError):
        host.get_vocoder({"vocoder": {"model": "MelGAN", "speaker": "universal"}}, "cpu", root=str(tmp_path))
    with pytest.raises(FileNotFoundError):
        host.get_vocoder({"vocoder": {"model": "HiFi-GAN", "speaker": "LJSpeech"}}, "cpu", root=str(tmp_path))


def test_collate_groups_layout():
    """host.collate_groups (round 4): bucket groups of a ragged shard -> one padded text batch per attention class (L <= 192 / longer), with each
    utterance's OWN group's padded length in pad_lens and the members table that maps the batch back to the groups."""
    from cmtts_amd import host
    gen = torch.Generator().manual_seed(0)
    groups = []
    for n, L, bucket in ((3, 10, 64), (2, 200, 1024), (4, 43, 256)):
        tx = torch.randint(1, 100, (n, L), generator=gen)
        ln = torch.randint(1, L + 1, (n,), generator=gen)
        groups.append((tx, ln, torch.randn(n, 512, generator=gen), None, bucket, torch.arange(n)))
    coll = host.collate_groups(groups, "cpu")
    assert coll.n == [3, 2, 4] and [b for _, b in coll.groups] == [64, 1024, 256]
    short, long_ = coll.batches
    assert tuple(short["texts"].shape) == (7, 43) and tuple(long_["texts"].shape) == (2, 200)
    assert short["pad_lens"].tolist() == [10] * 3 + [43] * 4 and long_["pad_lens"].tolist() == [200, 200]
    assert short["members"] == [(0, 0, 3, 10), (2, 3, 4, 43)] and long_["members"] == [(1, 0, 2, 200)]
    assert torch.equal(short["texts"][:3, :10], groups[0][0]) and not short["texts"][:3, 10:].any()       # padded with the pad symbol 0
    assert torch.equal(short["texts"][3:], groups[2][0])
    assert torch.equal(short["src_lens"], torch.cat([groups[0][1], groups[2][1]]))
    assert tuple(short["spk"].shape) == (7, 512) and short["speakers"].tolist() == [0, 1, 2, 0, 1, 2, 3]


def test_cond_factors_bind_to_one_tensor():
    """host.CondFactors: the conditioner factors apply to the conditioning tensor they were made with, unmodified — anything else takes the dense GEMM."""
    from cmtts_amd import host
    cond = torch.zeros(2, 256, 8)
    f = host.CondFactors(torch.zeros(2, 5120, 4), 4, 3, torch.zeros(2, 8, dtype=torch.int64), torch.zeros(2, 8, dtype=torch.int64), cond)
    assert f.matches(cond) and f.matches(cond.contiguous())
    assert not f.matches(cond.clone()) and not f.matches(cond[:1])
    cond.transpose(1, 2).add_(1.0)              # an in-place edit through a view bumps the shared version counter
    assert not f.matches(cond)
