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
