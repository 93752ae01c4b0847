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
    n_speaker: int = 0                # > 0: preprocess.yaml speaker_embedder "none" -> nn.Embedding(n_speaker, hidden) (model/cmtts.py:26-38)
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
    # multi-speaker with the embedding table instead of an external embedder (speaker_embedder: "none"; speakers.json of
    # the VCTK corpus lists 108 speakers — any n_speaker works, the table comes with the checkpoint)
    "VCTK_table": CMTTSConfig(name="VCTK_table", multi_speaker=True, use_uv=True, n_speaker=108),
}


def get_config(name: str) -> CMTTSConfig:
    return VARIANTS[name]


def config_from_reference(preprocess_config: dict, model_config: dict, train_config: dict = None, n_symbols: int = 361,
                          n_speaker: int = None, energy_min: float = None, energy_max: float = None) -> CMTTSConfig:
    """CMTTSConfig from the reference's three YAML dicts (config/<dataset>/{preprocess,model,train}.yaml), the way
    DurationPitchSpeakerNet / VarianceAdaptor / Denoiser read them (model/cmtts.py:15-42, model/modules.py:107-130,
    167-256,562-598; script_util.py:56-76 for the `cm` block).  Inputs the reference takes from files next to the
    checkpoint are arguments here: `n_speaker` (len(speakers.json), only for speaker_embedder "none"; read from
    <preprocessed_path>/speakers.json when omitted) and the energy range (stats.json "energy"[:2]; only recorded — the
    bucket boundaries themselves come with the state dict as variance_adaptor.energy_bins)."""
    import json
    import os
    pre, tr, vp, ve = preprocess_config["preprocessing"], model_config["transformer"], model_config["variance_predictor"], \
        model_config["variance_embedding"]
    pitch = pre["pitch"]
    if pitch["pitch_type"] != "cwt":
        raise NotImplementedError("only pitch_type 'cwt' (what every CM-TTS config uses) is on the hot path")
    if pre["energy"]["feature"] != "phoneme_level":
        raise NotImplementedError("only phoneme-level energy (what every CM-TTS config uses) is on the hot path")
    multi = bool(model_config["multi_speaker"])
    table = multi and pre.get("speaker_embedder", "none") == "none"
    if table and n_speaker is None:
        with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")) as f:
            n_speaker = len(json.load(f))
    cm = (train_config or {}).get("cm", {})
    kw = dict(
        name=str(preprocess_config.get("dataset", "custom")), n_symbols=n_symbols, hidden=tr["encoder_hidden"],
        enc_layers=tr["encoder_layer"], enc_heads=tr["encoder_head"], ffn_kernel=tr["ffn_kernel_size"],
        pred_filter=vp["filter_size"], pred_layers=vp["predictor_layers"], pred_kernel=vp["predictor_kernel"],
        dur_layers=vp["dur_predictor_layers"], dur_kernel=vp["dur_predictor_kernel"], cwt_hidden=vp["cwt_hidden_size"],
        cwt_std_scale=float(vp["cwt_std_scale"]), pitch_bins=ve["pitch_n_bins"], energy_bins=ve["energy_n_bins"],
        use_uv=bool(pitch["use_uv"]), pitch_norm_eps=float(pitch["pitch_norm_eps"]), multi_speaker=multi,
        external_speaker_dim=int(model_config.get("external_speaker_dim", 512)), n_speaker=int(n_speaker) if table else 0,
        n_mels=pre["mel"]["n_mel_channels"], res_layers=model_config["denoiser"]["residual_layers"],
        res_channels=model_config["denoiser"]["residual_channels"],
        sigma_min=float(cm.get("sigma_min", 0.002)), sigma_max=float(cm.get("sigma_max", 80.0)),
        sigma_data=float(cm.get("sigma_data", 0.5)), rho=float(cm.get("rho", 7.0)),
        hop_length=pre["stft"]["hop_length"], sampling_rate=pre["audio"]["sampling_rate"],
        max_wav_value=float(pre["audio"]["max_wav_value"]))
    if energy_min is not None:
        kw["energy_min"] = float(energy_min)
    if energy_max is not None:
        kw["energy_max"] = float(energy_max)
    return CMTTSConfig(**kw)
