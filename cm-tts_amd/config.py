"""Model hyper-parameters of the CM-TTS inference hot path.

Values restate the reference YAMLs (config/{LJSpeech,VCTK,LibriTTS}/{model,preprocess,train}.yaml)
and hifigan/config.json; nothing here is read from /root/reference at run time.
"""
from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass(frozen=True)
class CMTTSConfig:
    name: str = "LJSpeech"
    # text encoder: config/*/model.yaml:1-12
    n_symbols: int = 361              # len(text.symbols)+1, model/modules.py:124-126
    hidden: int = 256
    enc_layers: int = 4
    enc_heads: int = 2
    ffn_kernel: int = 9
    # variance predictors: config/*/model.yaml:34-45
    pred_filter: int = 256
    pred_layers: int = 2
    pred_kernel: int = 5
    dur_layers: int = 2
    dur_kernel: int = 3
    cwt_hidden: int = 128
    cwt_std_scale: float = 0.8
    pitch_bins: int = 300
    energy_bins: int = 256
    # preprocess.yaml pitch/energy block
    use_uv: bool = True               # LibriTTS: False (config/LibriTTS/preprocess.yaml:34)
    pitch_norm_eps: float = 1e-9
    energy_min: float = -1.5          # stats.json "energy"[:2]; fabricated (blob missing)
    energy_max: float = 8.0
    # speakers: config/VCTK/model.yaml:1-2,55
    multi_speaker: bool = False
    external_speaker_dim: int = 512
    # denoiser: config/*/model.yaml:14-18
    n_mels: int = 80
    res_layers: int = 20
    res_channels: int = 256
    # consistency sampler: config/*/train.yaml cm: sigma_min/sigma_max, script_util.py:67
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    sigma_data: float = 0.5
    rho: float = 7.0
    # audio
    hop_length: int = 256
    sampling_rate: int = 22050
    max_wav_value: float = 32768.0

    @property
    def cwt_out(self) -> int:
        return 11 if self.use_uv else 10

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class HifiGanConfig:
    """hifigan/config.json:11-15 (V1 generator); one architecture for both vocoder checkpoints."""
    num_mels: int = 80
    upsample_rates: Tuple[int, ...] = (8, 8, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    lrelu_slope: float = 0.1
    final_lrelu_slope: float = 0.01   # F.leaky_relu default, hifigan/models.py:161

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h


VARIANTS = {
    "LJSpeech": CMTTSConfig(name="LJSpeech", multi_speaker=False, use_uv=True),
    "VCTK": CMTTSConfig(name="VCTK", multi_speaker=True, use_uv=True),
    "LibriTTS": CMTTSConfig(name="LibriTTS", multi_speaker=True, use_uv=False),
}


def get_config(name: str) -> CMTTSConfig:
    return VARIANTS[name]
