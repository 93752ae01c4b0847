"""cmtts_amd — MI355X-native CM-TTS inference hot path (package directory: ``cm-tts_amd/``).

Hand-written HIP kernels for gfx950 behind a C ABI (``include/cmtts_hip.h``,
``cm-tts_amd/csrc/``), with a Python host layer that mirrors the reference's call surface
(``synthesize.py`` / ``CMTotalTTS`` / ``CMDenoiserTTS.forward`` / ``hifigan.Generator.forward``).
"""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; the library's side streams plus a
# caller's own streams (bucket groups) then share queues and serialise falsely (configs[3] shard: 27.1 ms with 4 queues,
# 23.7 ms with 8).  Read by the HIP runtime when it initialises, i.e. at the first device call: set it here unless the
# user already chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import CMTTSConfig, HifiGanConfig, VARIANTS, get_config  # noqa: F401

__version__ = "0.1.0"
