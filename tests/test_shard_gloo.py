"""N>1 path on CPU: world_size-2 gloo run of the one collective on the inference path
(cmtts_amd.shard.allgather_mels) plus the host-side shard arithmetic."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cmtts_amd
from cmtts_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Bl, T, M = 3, 16, 80
    g = torch.Generator().manual_seed(100 + rank)
    mel = torch.randn(Bl, T, M, generator=g)
    mel_len = torch.tensor([T - rank, 5 + rank, 9], dtype=torch.int64)
    all_mel, all_len = shard.allgather_mels(mel, mel_len)
    # the pipelined form (bench.py): issue, overwrite the producer's buffer as the next batch would, then wait
    src = mel.clone()
    pend = shard.allgather_mels_async(src, mel_len)
    src.zero_()
    a_mel, a_len = pend.wait()
    assert torch.equal(a_mel, all_mel) and torch.equal(a_len, all_len)
    # end-to-end wav jobs (BASELINE.json configs[4]): the int16 PCM block + sample counts in ONE all-gather of bytes
    N = 517                                           # not a multiple of 4: the packed row pads before the int64 count
    gp = torch.Generator().manual_seed(500 + rank)
    pcm = torch.randint(-32768, 32768, (Bl, N), generator=gp, dtype=torch.int32).to(torch.int16)
    wav_len = torch.tensor([N, 256 * (rank + 1), 1], dtype=torch.int64)
    all_pcm, all_wl = shard.allgather_pcm(pcm, wav_len)
    assert all_pcm.dtype == torch.int16 and all_pcm.shape == (world * Bl, N) and all_wl.dtype == torch.int64
    for r in range(world):
        gr = torch.Generator().manual_seed(500 + r)
        ref = torch.randint(-32768, 32768, (Bl, N), generator=gr, dtype=torch.int32).to(torch.int16)
        assert torch.equal(all_pcm[r * Bl:(r + 1) * Bl], ref)
        assert all_wl[r * Bl:(r + 1) * Bl].tolist() == [N, 256 * (r + 1), 1]
    utts = shard.restore_pcm_order(all_pcm, all_wl, [[4, 0, -1], [1, 3, 2]], 5)      # rank-major deal, -1 = filler row
    assert [u.numel() for u in utts] == [256, N, 1, 512, N]
    q.put((rank, all_mel.numpy(), all_len.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_mels_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect_mel, expect_len = [], []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        expect_mel.append(torch.randn(3, 16, 80, generator=g).numpy())
        expect_len.append(np.asarray([16 - r, 5 + r, 9]))
    expect_mel, expect_len = np.concatenate(expect_mel), np.concatenate(expect_len)
    for _, m, l in res:                       # every rank holds the full, rank-ordered collation
        assert np.array_equal(m, expect_mel)  # bit-exact: a gather moves bytes
        assert np.array_equal(l, expect_len)


def test_shard_arithmetic():
    for n, w in [(256, 8), (10, 4), (3, 8), (128, 1)]:
        spans = [shard.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    lens = [900, 100, 500, 510, 20, 1000, 30, 700]
    deal = shard.deal_by_length(lens, 4)
    assert sorted(i for d in deal for i in d) == list(range(8))
    assert all(len(d) == 2 for d in deal)
    assert [shard.frame_bucket(t) for t in (1, 256, 257, 1024)] == [256, 256, 512, 1024]
    mel = torch.arange(2 * 4 * 80, dtype=torch.float32).reshape(2, 4, 80)
    m2, l2 = shard.unpack_mels(shard.pack_mels(mel, torch.tensor([4, 3])), 4, 80)
    assert torch.equal(m2, mel) and l2.tolist() == [4, 3]
    for N in (8, 9, 10, 11, 1024):
        pcm = (torch.arange(3 * N, dtype=torch.int32).reshape(3, N) * 37 - 20000).to(torch.int16)
        wl = torch.tensor([N, 2 ** 40 + 5, 0], dtype=torch.int64)          # the count travels as a full int64
        buf = shard.pack_pcm(pcm, wl)
        assert buf.shape == (3, (N + 3) // 4 * 4 + 4) and buf.dtype == torch.int16
        p2, w2 = shard.unpack_pcm(buf, N)
        assert torch.equal(p2, pcm) and torch.equal(w2, wl)
    p1, w1 = shard.allgather_pcm(pcm, wl)                                    # no process group: identity
    assert torch.equal(p1, pcm) and torch.equal(w1, wl)


FRAMES = [900, 100, 500, 510, 20, 1000, 30, 700, 255, 257, 600]     # 11 utterances: 3 / 3 / 2 / 3 per bucket


def _fake_synth(idx, bucket):
    """Stand-in for text -> mel on one rank: mel[t, m] = 1000 * utterance + t + m / 100 for t < n_frames, 0 after."""
    n = FRAMES[idx]
    mel = torch.zeros(bucket, 80)
    mel[:n] = 1000.0 * idx + torch.arange(n, dtype=torch.float32)[:, None] + torch.arange(80, dtype=torch.float32)[None, :] / 100
    return mel, n


def _worker_plan(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = shard.plan_shards(FRAMES, world)
    gathered = {}
    for bucket, ranks in plan.items():
        mine = ranks[rank]
        mels, lens = zip(*[_fake_synth(i if i >= 0 else ranks[rank][0], bucket) for i in mine])
        gathered[bucket] = shard.allgather_mels_async(torch.stack(mels), torch.tensor(lens, dtype=torch.int64)).wait()
    out = shard.restore_order(gathered, plan, len(FRAMES))
    # the same shard through ONE all-gather for all buckets (north_star: "a single RCCL all-gather")
    local = {}
    for bucket, ranks in plan.items():
        mels, lens = zip(*[_fake_synth(i if i >= 0 else ranks[rank][0], bucket) for i in ranks[rank]])
        local[bucket] = (torch.stack(mels), torch.tensor(lens, dtype=torch.int64))
    one = shard.allgather_buckets(local)
    for b in plan:
        assert torch.equal(one[b][0], gathered[b][0]) and torch.equal(one[b][1], gathered[b][1])
    out1 = shard.restore_order(one, plan, len(FRAMES))
    assert all(torch.equal(a, b) for a, b in zip(out, out1))
    q.put((rank, [o.numpy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_shard_plan_world2():
    """The configs[3] flow on 2 ranks: bucket, deal, synthesize the local share per bucket, one all-gather per bucket,
    restore the original order — every rank ends with every utterance's mel, trimmed to its length."""
    world, port = 2, _free_port()
    plan = shard.plan_shards(FRAMES, world)
    assert sorted(plan) == [256, 512, 768, 1024]
    assert sorted(i for ranks in plan.values() for r in ranks for i in r if i >= 0) == list(range(len(FRAMES)))
    assert all(len({len(r) for r in ranks}) == 1 for ranks in plan.values())          # equal blocks per rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_plan, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, mels in res:
        for i, m in enumerate(mels):
            ref, n = _fake_synth(i, shard.frame_bucket(FRAMES[i]))
            assert m.shape == (n, 80) and np.array_equal(m, ref[:n].numpy())


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] at its own size: 256 ragged utterances over 8 ranks, and bench.py's N-GPU control flow


def _cfg3_lengths(world=8, per_bucket=8, dur=6):
    """bench.ragged_groups' recipe: per static frame bucket, per_bucket x world utterances, lengths ~ U[0.5, 1] x bucket that stay in
    THEIR bucket, durations forced to `dur` frames per phoneme."""
    rs = np.random.RandomState(40)
    frames = []
    for bi, bucket in enumerate(shard.FRAME_BUCKETS):
        Lmax = bucket // dur
        prev = shard.FRAME_BUCKETS[bi - 1] // dur if bi else 0
        ln = np.maximum((rs.uniform(0.5, 1.0, size=per_bucket * world) * Lmax).astype(np.int64), prev + 1)
        frames += [int(v) * dur for v in ln]
    return frames


def _mel_of(idx, n, bucket, M):
    """Deterministic stand-in for utterance idx's mel: distinct per (utterance, frame, mel bin), zero beyond its length."""
    mel = torch.zeros(bucket, M)
    mel[:n] = (idx * 4099 % 9973) + torch.arange(n, dtype=torch.float32)[:, None] * 0.25 + torch.arange(M, dtype=torch.float32)[None, :] / 128
    return mel


def _worker_cfg3(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _cfg3_lengths(world)
    plan = shard.plan_shards(frames, world)
    M = 80
    local = {}
    for bucket, ranks in plan.items():
        mine = ranks[rank]
        assert len(mine) == 8 and all(i >= 0 for i in mine)          # 32 per rank in four buckets, no filler rows at this size
        mels = torch.stack([_mel_of(i, frames[i], bucket, M) for i in mine])
        local[bucket] = (mels, torch.tensor([frames[i] for i in mine], dtype=torch.int64))
    assert sum(v[0].shape[0] for v in local.values()) == 32
    got = shard.allgather_buckets(local)                                # ONE all-gather for the whole shard
    for bucket, (mel, ln) in got.items():
        assert mel.shape == (world * 8, bucket, M) and ln.shape == (world * 8,)
        assert torch.equal(mel[rank * 8:(rank + 1) * 8], local[bucket][0])     # this rank's block sits at its rank offset
    utts = shard.restore_order(got, plan, len(frames))
    ok = all(u.shape == (frames[i], M) and torch.equal(u, _mel_of(i, frames[i], shard.frame_bucket(frames[i]), M)[:frames[i]])
             for i, u in enumerate(utts))
    nbytes = sum(v[0].numel() * 4 + v[1].numel() * 8 for v in local.values())
    q.put((rank, ok, len(utts), nbytes))
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_recipe_world8():
    """256 utterances with bench.py's configs[3] length recipe -> plan_shards -> 32 per rank in four buckets of 8 (equal blocks, no
    filler) -> allgather_buckets (one collective) -> restore_order: every rank ends with all 256 mels, trimmed, in the original
    order, bit for bit.  10.5 MB per rank per gather at the full mel width — what DESIGN §5 prices the xGMI step on."""
    world, port = 8, _free_port()
    frames = _cfg3_lengths(world)
    assert len(frames) == 256
    plan = shard.plan_shards(frames, world)
    assert sorted(plan) == [256, 512, 768, 1024]
    assert all(len(ranks) == world and all(len(r) == 8 for r in ranks) for ranks in plan.values())
    assert sorted(i for ranks in plan.values() for r in ranks for i in r) == list(range(256))
    # dealt by length: the ranks' frame totals of a bucket differ by less than one utterance of that bucket
    for bucket, ranks in plan.items():
        tot = [sum(frames[i] for i in r) for r in ranks]
        assert max(tot) - min(tot) <= bucket, (bucket, tot)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cfg3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world)) and all(r[1] and r[2] == 256 for r in res)
    assert all(r[3] == 8 * (256 + 512 + 768 + 1024) * 80 * 4 + 32 * 8 for r in res)          # 6.6 MB of mel per rank at this mix (10.5 MB when all 32 are 1024-frame rows)


class _StandInHost:
    """CPU stand-in for cmtts_amd.host with the calls bench.multi_gpu_extras makes (same shapes, dtypes and collation contract; values
    are cheap deterministic functions of the inputs).  What the rehearsal pins is the CONTROL FLOW every rank must agree on: the order
    of collectives and barriers, block shapes, the restored orders and the JSON fields — not kernels."""

    class CMTotalTTS:
        def __init__(self, cfg, device):
            self.config, self.device = cfg, device

        def load_state_dict(self, sd):
            return self

        def set_precision(self, mode):
            self.mode = mode

        def duration_pitch_energy_net(self, speakers, texts, lens, spker_embeds=None, max_mel_len=None):
            B = texts.shape[0]
            return {"cond_ct": torch.zeros(B, 4, max_mel_len), "speaker_emb": None, "mel_lens": lens * 6, "cond_factors": None}

    class Generator:
        def __init__(self, hcfg, device):
            self.hop = 256

        def load_state_dict(self, sd):
            return self

    class BucketedSynthesizer:
        def __init__(self, model, n_steps, n_streams=4):
            self.model = model

        def run(self, coll):
            outs = []
            for (texts, lens, spk, noise, bucket) in coll:
                n = texts.shape[0]
                ln = lens * 6
                mel = torch.zeros(n, bucket, 80)
                for r in range(n):
                    mel[r, :int(ln[r])] = float(texts[r, 0]) + torch.arange(int(ln[r]), dtype=torch.float32)[:, None] / 64
                outs.append((mel, ln))
            return outs

    @staticmethod
    def collate_groups(groups, device):
        return list(groups)

    @staticmethod
    def sample_with_cond(model, cond_ct, spk, n_steps, nz, factors=None):
        return nz[0, :, 0] * 0.0 + 1.0

    @staticmethod
    def vocoder_infer_device(mels, voc):
        B, _, T = mels.shape
        return (torch.arange(B * T * 256, dtype=torch.int64).reshape(B, T * 256) % 30011 - 15000).to(torch.int16)

    @staticmethod
    def check_async_error():
        return None


def _worker_extras(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import bench
    torch.cuda.synchronize = lambda *a, **k: None           # this process has no GPU: the stand-ins compute on the host
    bench.host = _StandInHost
    bench.REDUCE_DEVICE = "cpu"
    bench.synth_cmtts_state_dict = lambda *a, **k: {}
    bench.synth_hifigan_state_dict = lambda *a, **k: {}
    calls = []
    state = {}

    def step(n):
        calls.append(n)

    def timed_w(fn, k, warm):
        return bench.timed(fn, k, warm, world)

    args = argparse.Namespace(steps=4)
    extras = bench.multi_gpu_extras(args, None, None, step, timed_w, state, frames_rank=100, audio_s=1.0, rank=rank, world=world,
                                    device=torch.device("cpu"), gather=True)
    q.put((rank, extras, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_gpu_control_flow_world2():
    """bench.py's N-GPU extras (T = 1 / 2 rates, configs[3] through plan_shards + one all-gather of all buckets + restore_order,
    configs[4] through the PCM all-gather) on two gloo ranks with a CPU stand-in synthesizer: the function's own assertions (restored
    lengths = the plan's, every rank's block at its rank offset, PCM counts) hold on both ranks, both reach every barrier, and the
    JSON fields are whole-job aggregates."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_extras, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, extras, calls in res:
        assert set(calls) == {1, 2}
        assert {"frames_per_s_T1", "rtf_mel_only_T1", "frames_per_s_T2", "configs3_ragged_bucketed", "configs4_end_to_end_wav"} <= set(extras)
        c3, c4 = extras["configs3_ragged_bucketed"], extras["configs4_end_to_end_wav"]
        assert c3["valid_frames"] == sum(_cfg3_lengths(world)) and c3["frames_per_s"] > 0
        assert c4["frames_per_s"] > 0 and "16 utterances" in c4["workload"]
    assert res[0][1]["configs3_ragged_bucketed"]["valid_frames"] == res[1][1]["configs3_ragged_bucketed"]["valid_frames"]
