"""cmtts_amd — MI355X-native CM-TTS inference hot path (package directory: ``cm-tts_amd/``).

Hand-written HIP kernels for gfx950 behind a C ABI (``include/cmtts_hip.h``,
``cm-tts_amd/csrc/``), with a Python host layer that mirrors the reference's call surface
(``synthesize.py`` / ``CMTotalTTS`` / ``CMDenoiserTTS.forward`` / ``hifigan.Generator.forward``).
"""
from .config import CMTTSConfig, HifiGanConfig, VARIANTS, get_config  # noqa: F401

__version__ = "0.1.0"
