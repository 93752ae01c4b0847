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
