"""The N>1 call sequence of bench.py on ONE MI355X: a 1-rank RCCL ("nccl") group, the asynchronous mel all-gather of
batch i overlapping the text side of batch i+1, then the full-chip persistent denoiser launch — 20 iterations, plain and
cooperative launch.  What it proves without an 8-GPU node: RCCL initialises next to the library, its kernels and the
persistent kernel (which needs every CU resident) coexist in the order the bench issues them, no neighbour wait times
out (cmtts_poll_error after every synchronise), and the collated block is bit-identical to the producer's mel."""
import os

import numpy as np
import pytest
import torch

import cmtts_amd
from cmtts_amd import _lib, shard
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("cooperative", [0, 1])
def test_async_gather_then_persistent_launch(nccl_group, cooperative):
    from cmtts_amd import host
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
    B, L, T, n_steps = 32, 85, 512, 4
    rs = np.random.RandomState(0)
    batches = [torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).to(DEV) for _ in range(2)]
    lens = torch.full((B,), L, dtype=torch.int64, device=DEV)
    noise = torch.randn(n_steps + 1, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(3)).to(DEV)

    def text_to_mel(texts):
        out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
        return out, host.sample_with_cond(model, out["cond_ct"], None, n_steps, noise)

    prev_p = lib.cmtts_set_persistent_denoiser(2)
    prev_c = lib.cmtts_set_option(b"cooperative_launch", cooperative)
    try:
        ref = []
        for tx in batches:      # no collective in flight
            _, mel = text_to_mel(tx)
            host.synchronize()
            ref.append(mel.clone())
        pending, produced = None, []
        for it in range(20):
            tx = batches[it % 2]
            out = model.duration_pitch_energy_net(None, tx, lens, max_mel_len=T)     # overlaps the gather of batch it-1
            if pending is not None:      # RCCL's kernels off the GPU before the launch that needs every CU
                g_mel, g_len = pending.wait()
                assert torch.equal(g_mel, produced[-1]) and (g_len == 6 * L).all()
            mel = host.sample_with_cond(model, out["cond_ct"], None, n_steps, noise)
            pending = shard.allgather_mels_async(mel, out["mel_lens"], force=True)
            produced.append(mel.clone())
            assert lib.cmtts_poll_error() == 0
        g_mel, g_len = pending.wait()
        host.synchronize()               # raises if any persistent launch timed out
        assert torch.equal(g_mel, produced[-1])
        for it, mel in enumerate(produced):
            assert torch.equal(mel, ref[it % 2]), f"iteration {it}: mel differs from the run without a collective"
        # the single-buffer ragged plan through the same group (configs[3]: all buckets in ONE all-gather)
        plan = shard.plan_shards([100, 300, 260, 700, 90, 1000], 1)
        mels = {b: (torch.randn(len(r[0]), b, cfg.n_mels, device=DEV), torch.randint(1, b, (len(r[0]),), device=DEV))
                for b, r in plan.items()}
        gathered = shard.allgather_buckets(mels, force=True)
        for b in plan:
            assert torch.equal(gathered[b][0], mels[b][0]) and torch.equal(gathered[b][1], mels[b][1])
    finally:
        lib.cmtts_set_option(b"cooperative_launch", prev_c)
        lib.cmtts_set_persistent_denoiser(prev_p)


def test_c_abi_allgather_on_a_one_rank_communicator():
    """cmtts_comm_unique_id / cmtts_comm_init_rank / cmtts_allgather_mels / cmtts_comm_destroy (include/cmtts_hip.h): the
    collective a non-Python host calls.  A 1-rank RCCL communicator runs the real ncclAllGather; the packed buffer carries
    mel_len next to the mel block; the result must be the input, bit for bit, and the no-communicator form (world = 1,
    comm = NULL) must agree."""
    import ctypes as C
    lib = _lib.load()
    torch.cuda.set_device(0)
    uid = (C.c_char * 128)()
    _lib.check(lib.cmtts_comm_unique_id(C.cast(uid, C.c_void_p)))
    comm = C.c_void_p()
    _lib.check(lib.cmtts_comm_init_rank(C.byref(comm), 1, 0, C.cast(uid, C.c_void_p)))
    try:
        Bl, T, M = 5, 96, 80
        g = torch.Generator().manual_seed(4)
        mel = torch.randn(Bl, T, M, generator=g).to(DEV)
        mel_len = torch.tensor([96, 1, 50, 77, 13], dtype=torch.int64, device=DEV)
        for use_comm in (True, False):
            out_mel = torch.full((Bl, T, M), float("nan"), device=DEV)
            out_len = torch.zeros(Bl, dtype=torch.int64, device=DEV)
            nb = lib.cmtts_allgather_workspace_bytes(1, Bl, T, M)
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            _lib.check(lib.cmtts_allgather_mels(comm if use_comm else None, 1, mel.data_ptr(), mel_len.data_ptr(), Bl, T, M,
                                                out_mel.data_ptr(), out_len.data_ptr(), ws.data_ptr(), nb,
                                                torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert torch.equal(out_mel, mel) and torch.equal(out_len, mel_len)
        assert lib.cmtts_allgather_mels(None, 2, mel.data_ptr(), mel_len.data_ptr(), Bl, T, M, mel.data_ptr(), mel_len.data_ptr(),
                                        mel.data_ptr(), 1 << 30, None) < 0            # world > 1 needs a communicator
    finally:
        _lib.check(lib.cmtts_comm_destroy(comm))


def test_pcm_gather_torch_and_c_abi_one_rank(nccl_group):
    """SURVEY.md §8e, BASELINE.json configs[4]: the int16 wav collation of end-to-end jobs.  shard.allgather_pcm on a 1-rank
    RCCL group (force) and cmtts_allgather_pcm on a 1-rank communicator / without one must both return the producer's PCM
    and sample counts bit for bit, including a row length that is not a multiple of 4 (the count is packed behind the row)."""
    import ctypes as C
    lib = _lib.load()
    for N in (4 * 256, 1021):
        Bl = 3
        pcm = torch.randint(-32768, 32768, (Bl, N), dtype=torch.int32, generator=torch.Generator().manual_seed(N)).to(torch.int16).to(DEV)
        wav_len = torch.tensor([N, 256, 1], dtype=torch.int64, device=DEV)
        g_pcm, g_len = shard.allgather_pcm(pcm, wav_len, force=True)
        assert torch.equal(g_pcm, pcm) and torch.equal(g_len, wav_len)
        uid = (C.c_char * 128)()
        _lib.check(lib.cmtts_comm_unique_id(C.cast(uid, C.c_void_p)))
        comm = C.c_void_p()
        _lib.check(lib.cmtts_comm_init_rank(C.byref(comm), 1, 0, C.cast(uid, C.c_void_p)))
        try:
            for use_comm in (True, False):
                out_pcm = torch.full((Bl, N), 77, dtype=torch.int16, device=DEV)
                out_len = torch.zeros(Bl, dtype=torch.int64, device=DEV)
                nb = lib.cmtts_allgather_pcm_workspace_bytes(1, Bl, N)
                ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
                _lib.check(lib.cmtts_allgather_pcm(comm if use_comm else None, 1, pcm.data_ptr(), wav_len.data_ptr(), Bl, N,
                                                   out_pcm.data_ptr(), out_len.data_ptr(), ws.data_ptr(), nb,
                                                   torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                assert torch.equal(out_pcm, pcm) and torch.equal(out_len, wav_len)
            assert lib.cmtts_allgather_pcm(None, 2, pcm.data_ptr(), wav_len.data_ptr(), Bl, N, pcm.data_ptr(), wav_len.data_ptr(),
                                           pcm.data_ptr(), 1 << 30, None) < 0            # world > 1 needs a communicator
        finally:
            _lib.check(lib.cmtts_comm_destroy(comm))


def test_cooperative_launch_is_automatic_with_a_process_group(nccl_group):
    """with a process group in the process the persistent denoiser is launched cooperatively by default
    (option value 2 = automatic); 0 / 1 force it; the previous value comes back."""
    from cmtts_amd import host
    lib = _lib.load()
    cfg = get_config("LJSpeech")
    host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))._require()
    assert lib.cmtts_set_option(b"process_group", -1) == 1           # the host layer told the library (torch.distributed is up)
    prev = lib.cmtts_set_option(b"cooperative_launch", 2)
    try:
        assert lib.cmtts_set_option(b"cooperative_launch", 0) == 2
        assert lib.cmtts_set_option(b"cooperative_launch", 1) == 0
        assert lib.cmtts_set_option(b"cooperative_launch", 2) == 1
    finally:
        lib.cmtts_set_option(b"cooperative_launch", prev)
