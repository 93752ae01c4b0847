"""Binding of libcmtts_hip.so (the C ABI in include/cmtts_hip.h): ctypes by default; cffi in ABI mode
(BASELINE.json north_star: "a thin C-ABI cffi layer"; declarations parsed from the header itself) with ``CMTTS_FFI=cffi`` —
``cffi`` is not installed in this image, so ctypes is what runs here and what the tests exercise, and the
cffi adapter stays opt-in until it has run somewhere; both bind the same symbols with the same signatures behind the same call sites.
There is NO compute fallback: if the shared library is missing or a symbol is absent the import of the compute path
fails loudly.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMTTS_LIB") or os.path.join(_HERE, "libcmtts_hip.so")     # CMTTS_LIB: an experimental build of the same ABI (tools/)


class CMTTSConfigStruct(C.Structure):
    """struct cmtts_config (include/cmtts_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_symbols", "hidden", "enc_layers", "enc_heads", "ffn_kernel",
        "pred_filter", "pred_layers", "pred_kernel", "dur_layers", "dur_kernel", "cwt_hidden",
        "pitch_bins", "energy_bins", "use_uv", "multi_speaker", "external_speaker_dim", "n_speaker",
        "n_mels", "res_layers", "res_channels")] + [(n, C.c_float) for n in (
            "cwt_std_scale", "pitch_norm_eps", "sigma_min", "sigma_max", "sigma_data", "rho")]


class VarianceControlsStruct(C.Structure):
    """struct cmtts_variance_controls (include/cmtts_hip.h): device pointers as void*."""
    _fields_ = [("p_control", C.c_float), ("e_control", C.c_float), ("d_target", C.c_void_p), ("e_target", C.c_void_p),
                ("cwt_spec", C.c_void_p), ("f0_mean", C.c_void_p), ("f0_std", C.c_void_p), ("uv", C.c_void_p)]


class SampleGroupStruct(C.Structure):
    """struct cmtts_sample_group (include/cmtts_hip.h)."""
    _fields_ = [("noise", C.c_void_p), ("cond_ct", C.c_void_p), ("speaker_emb", C.c_void_p), ("B", C.c_int32), ("T", C.c_int32),
                ("active_frames", C.c_void_p), ("mel", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("cond_p1", C.c_void_p), ("p1_ld", C.c_int32), ("L", C.c_int32), ("mel2ph", C.c_void_p), ("p_idx", C.c_void_p)]


_vp, _i, _f, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64

# name -> (restype, argtypes); must list every symbol include/cmtts_hip.h declares
SIGNATURES = {
    "cmtts_last_error": (C.c_char_p, []),
    "cmtts_version": (C.c_char_p, []),
    "cmtts_abi_version": (_i, []),
    "cmtts_create": (_i, [C.POINTER(CMTTSConfigStruct), C.POINTER(_vp)]),
    "cmtts_set_tensor": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "cmtts_finalize": (_i, [_vp]),
    "cmtts_destroy": (None, [_vp]),
    "cmtts_text_workspace_bytes": (_sz, [_vp, _i, _i]),
    "cmtts_text_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_text_forward_ragged": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_set_variance_controls": (_i, [_vp, C.POINTER(VarianceControlsStruct)]),
    "cmtts_frame_workspace_bytes": (_sz, [_vp, _i, _i]),
    "cmtts_frame_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_frame_forward_sub": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_frame_forward_sub_t": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_length_regulate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cmtts_denoiser_workspace_bytes": (_sz, [_vp, _i, _i]),
    "cmtts_denoiser_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "cmtts_schedule": (_i, [_vp, _i, C.POINTER(_f), C.POINTER(_f)]),
    "cmtts_sample": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _sz, _vp]),
    "cmtts_sample_factored": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _sz, _vp, _vp, _i, _i, _vp, _vp]),
    "cmtts_sample_factored_t": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp, _vp, _sz, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "cmtts_sample_ragged": (_i, [_vp, C.POINTER(SampleGroupStruct), _i, _i, C.POINTER(_f), C.POINTER(_f), _i, _vp]),
    "cmtts_vocoder_create": (_i, [C.POINTER(_vp)]),
    "cmtts_vocoder_set_tensor": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "cmtts_vocoder_finalize": (_i, [_vp]),
    "cmtts_vocoder_destroy": (None, [_vp]),
    "cmtts_vocoder_workspace_bytes": (_sz, [_vp, _i, _i]),
    "cmtts_vocoder_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "cmtts_wav_to_int16": (_i, [_vp, _vp, _i64, _f, _vp]),
    "cmtts_profile_begin": (_i, [_i, _i]),
    "cmtts_set_fused_resblock": (_i, [_i]),
    "cmtts_set_persistent_denoiser": (_i, [_i]),
    "cmtts_poll_error": (_i, []),
    "cmtts_set_option": (_i, [C.c_char_p, _i]),
    "cmtts_model_set_option": (_i, [_vp, C.c_char_p, _i]),
    "cmtts_vocoder_set_option": (_i, [_vp, C.c_char_p, _i]),
    "cmtts_set_resblock_tile": (_i, [_i]),
    "cmtts_set_precision": (_i, [_vp, _i]),
    "cmtts_vocoder_set_precision": (_i, [_vp, _i]),
    "cmtts_set_debug_stamps": (_i, [_vp]),
    "cmtts_profile_end": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "cmtts_decoder_workspace_bytes": (_sz, [_vp, _i, _i]),
    "cmtts_decoder_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "cmtts_length_mask": (_i, [_vp, _vp, _i, _i, _vp]),
    "cmtts_comm_unique_id": (_i, [_vp]),
    "cmtts_comm_init_rank": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "cmtts_comm_destroy": (_i, [_vp]),
    "cmtts_allgather_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cmtts_allgather_mels": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_allgather_pcm_workspace_bytes": (_sz, [_i, _i, _i64]),
    "cmtts_allgather_pcm": (_i, [_vp, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "cmtts_transpose": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "cmtts_pack_conv_weight": (_i, [_vp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_i)]),
    "cmtts_free_device": (None, [_vp]),
    "cmtts_conv1d": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
}

ABI_VERSION = 5          # include/cmtts_hip.h: CMTTS_ABI_VERSION
_lib = None
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cmtts_hip.h")


def cdef_from_header(path=HEADER_PATH):
    """The header as cffi.FFI.cdef() takes it: comments, preprocessor lines and the extern "C" wrapper removed."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lines = [ln for ln in text.splitlines() if not ln.lstrip().startswith("#")]
    text = "\n".join(lines)
    text = text.replace('extern "C" {', "")
    text = re.sub(r"^\}\s*$", "", text, flags=re.M)          # the closing brace of extern "C" (struct bodies end in "} name;")
    return text


def address_of(arg):
    """Integer address of whatever the call sites pass for a pointer parameter: None, an int, a ctypes pointer value
    (c_void_p, cast), byref(obj), a ctypes array / structure, or an object with data_ptr()."""
    if arg is None:
        return 0
    if isinstance(arg, int):
        return arg
    if isinstance(arg, C.c_void_p):
        return arg.value or 0
    if hasattr(arg, "_obj"):                       # byref(x)
        return C.addressof(arg._obj)
    if isinstance(arg, (C.Array, C.Structure)):
        return C.addressof(arg)
    if isinstance(arg, C._Pointer):
        return C.cast(arg, C.c_void_p).value or 0
    if hasattr(arg, "data_ptr"):
        return int(arg.data_ptr())
    raise TypeError(f"cannot take the address of {type(arg).__name__}")


class _CffiLib:
    """libcmtts_hip.so through cffi (ABI mode: ffi.dlopen, no compiler), callable exactly like the ctypes object: pointer
    parameters accept what address_of() understands, `const char*` parameters take bytes, `const char*` results come back
    as bytes."""

    def __init__(self, path):
        import cffi
        self.ffi = cffi.FFI()
        self.ffi.cdef(cdef_from_header())
        self._c = self.ffi.dlopen(path)
        for name, (res, args) in SIGNATURES.items():
            setattr(self, name, self._wrap(name, res, args))

    def _wrap(self, name, res, args):
        ffi, fn = self.ffi, getattr(self._c, name)          # AttributeError if the .so lacks the symbol
        ctypes_ = ffi.typeof(fn).args
        assert len(ctypes_) == len(args), name

        def call(*a):
            conv = []
            for v, ct, at in zip(a, ctypes_, args):
                if at is C.c_char_p:
                    conv.append(ffi.NULL if v is None else v)
                elif ct.kind == "pointer":
                    conv.append(ffi.cast(ct, address_of(v)))
                else:
                    conv.append(v.value if hasattr(v, "value") else v)
            r = fn(*conv)
            if res is C.c_char_p:
                return None if r == ffi.NULL else ffi.string(r)
            return None if res is None else int(r)
        call.__name__ = name
        return call


def _load_ctypes():
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks the symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """dlopen libcmtts_hip.so and bind every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C cm-tts_amd/csrc). "
            "cmtts_amd has no CPU or PyTorch fallback.")
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and must bring them in FIRST: the library then binds to
    # that HIP runtime (same SONAME) and shares torch's device context and streams.  Loaded the other way round, the
    # process ends up with /opt/rocm's runtime under torch and "no ROCm-capable device is detected".
    import torch  # noqa: F401
    # ctypes is the default binding: it is the one every test in this image exercises.  The cffi ABI-mode binding is opt-in
    # (CMTTS_FFI=cffi) until a CI leg with cffi installed has run tests/test_cabi.py and a GPU smoke through it.
    want = os.environ.get("CMTTS_FFI", "ctypes")
    if want == "cffi":
        import cffi  # noqa: F401  (ImportError: the caller asked for a binding this environment does not have)
        lib = _CffiLib(LIB_PATH)
    elif want == "ctypes":
        lib = _load_ctypes()
    else:
        raise RuntimeError(f"CMTTS_FFI={want!r}: expected 'ctypes' or 'cffi'")
    got = lib.cmtts_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI revision {got}, this binding was written for {ABI_VERSION} "
                           "(include/cmtts_hip.h: CMTTS_ABI_VERSION): rebuild the library")
    _lib = lib
    # measurement hook (tools/, A/B runs of bench.py): CMTTS_INTERNAL="persist_wino=1,voc_wino=0" flips csrc/internal_hooks.h switches at load.
    # These switches change numerics, so a stray variable must not reach a production process: it is honoured only together with
    # CMTTS_INTERNAL_ENABLE=1, every entry is validated, an unknown name is an error, and what was applied is logged.
    spec = os.environ.get("CMTTS_INTERNAL", "")
    if spec:
        if os.environ.get("CMTTS_INTERNAL_ENABLE") != "1":
            print(f"cmtts_amd: ignoring CMTTS_INTERNAL={spec!r} (measurement hook: set CMTTS_INTERNAL_ENABLE=1 to apply it)", file=sys.stderr)
        else:
            for kv in filter(None, (e.strip() for e in spec.split(","))):
                k, sep, v = kv.partition("=")
                try:
                    val = int(v)
                except ValueError:
                    val = None
                if not sep or not k.strip() or val is None:
                    raise RuntimeError(f"CMTTS_INTERNAL: malformed entry {kv!r} (expected name=integer[,name=integer...])")
                if internal_set(k.strip(), val) < 0:
                    raise RuntimeError(f"CMTTS_INTERNAL: unknown switch {k.strip()!r} (csrc/internal_hooks.h)")
                print(f"cmtts_amd: CMTTS_INTERNAL applied {k.strip()}={val}", file=sys.stderr)
    return lib


_internal = None


def internal_set(name, value):
    """csrc/internal_hooks.h: flip one of the A/B switches between a fused kernel and the path it replaces (bitwise equal
    pairs; tests/ and tools/ only — not part of the C ABI, hence bound here and not in SIGNATURES).  Returns the previous value."""
    global _internal
    load()
    if _internal is None:
        _internal = C.CDLL(LIB_PATH).cmtts_internal_set
        _internal.restype, _internal.argtypes = _i, [C.c_char_p, _i]
    return _internal(name if isinstance(name, bytes) else name.encode(), int(value))


def backend():
    """"cffi" or "ctypes": which binding load() picked."""
    return "cffi" if isinstance(load(), _CffiLib) else "ctypes"


def check(rc):
    if rc != 0:
        msg = load().cmtts_last_error()
        raise RuntimeError(f"libcmtts_hip error {rc}: {msg.decode() if msg else '?'}")
