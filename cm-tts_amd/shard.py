"""Multi-GPU sharding of the inference path (new work: the reference's inference is single-process,
synthesize.py:32,43 — SURVEY.md §2a, §8e).

Independent units = utterances: no cross-utterance arithmetic exists anywhere on the path, so a
batch shards across ranks with no data-path collective; the only exchange is ONE all-gather that
collates the padded mel block (and, packed into the same buffer, each utterance's mel_len).
One process per GPU over ``torch.distributed`` — backend "nccl" is RCCL over xGMI on ROCm; "gloo"
for the CPU tests.

Because outputs depend on the padded (L_max, T_max) of the sub-batch (batch-padding
dependence is part of the semantics), shards pad to agreed static buckets rather than their local
maximum.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

FRAME_BUCKETS = (256, 512, 768, 1024)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the items this rank owns; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def deal_by_length(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-sorted round-robin deal: every rank gets a similar mix of long and short utterances,
    so padded work is balanced.  Returns per-rank index lists (original positions)."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    return [order[r::world] for r in range(world)]


def frame_bucket(n_frames: int, buckets: Sequence[int] = FRAME_BUCKETS) -> int:
    for b in buckets:
        if n_frames <= b:
            return int(b)
    raise ValueError(f"{n_frames} frames exceed the largest bucket {buckets[-1]}")


def plan_shards(n_frames: Sequence[int], world: int, buckets: Sequence[int] = FRAME_BUCKETS):
    """BASELINE.json configs[3] recipe: utterances go to the smallest static frame bucket that holds them; inside a
    bucket they are dealt round-robin by length over the ranks; every rank gets the SAME count per bucket (the all-gather
    needs equal blocks), short ranks are padded with -1 (the caller feeds any real utterance there and the restored list
    ignores it).  Returns {bucket: [[utterance index or -1] * per_rank for each rank]} — identical on every rank, no
    communication needed to agree on it."""
    plan = {}
    for b in buckets:
        ids = [i for i, t in enumerate(n_frames) if frame_bucket(int(t), buckets) == b]
        if not ids:
            continue
        deal = deal_by_length([n_frames[i] for i in ids], world)
        per_rank = max(len(d) for d in deal)
        plan[int(b)] = [[ids[j] for j in d] + [-1] * (per_rank - len(d)) for d in deal]
    return plan


def restore_order(gathered, plan, n_items: int):
    """gathered: {bucket: (mel [world*per_rank, bucket, M], mel_len [world*per_rank])} as returned by allgather_mels for
    every bucket of `plan`.  Returns the per-utterance mels [mel_len_i, M] in the ORIGINAL utterance order."""
    out = [None] * n_items
    for b, ranks in plan.items():
        mel, mel_len = gathered[b]
        flat = [i for r in ranks for i in r]
        assert mel.shape[0] == len(flat), (mel.shape, len(flat))
        for row, i in enumerate(flat):
            if i >= 0:
                out[i] = mel[row, : int(mel_len[row])]
    assert all(o is not None for o in out)
    return out


def pack_mels(mel: torch.Tensor, mel_len: torch.Tensor) -> torch.Tensor:
    """[Bl,T,M] fp32 + int64 [Bl] -> one fp32 buffer [Bl, T*M + 1] (mel_len < 2**24 is exact in fp32)."""
    Bl = mel.shape[0]
    buf = torch.empty(Bl, mel.shape[1] * mel.shape[2] + 1, dtype=torch.float32, device=mel.device)
    buf[:, :-1] = mel.reshape(Bl, -1)
    buf[:, -1] = mel_len.to(torch.float32)
    return buf


def unpack_mels(buf: torch.Tensor, T: int, M: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return buf[:, :-1].reshape(buf.shape[0], T, M), buf[:, -1].round().to(torch.int64)


def allgather_mels(mel: torch.Tensor, mel_len: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The path's single collective: every rank contributes [Bl,T,M] (same Bl, T on all ranks) and
    receives [world*Bl, T, M] plus the gathered mel_len, in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mel, mel_len
    world = dist.get_world_size(group)
    Bl, T, M = mel.shape
    buf = pack_mels(mel, mel_len)
    out = torch.empty(world * Bl, buf.shape[1], dtype=buf.dtype, device=buf.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, buf, group=group)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    return unpack_mels(out, T, M)


class PendingGather:
    """An all-gather of one batch's mels in flight (RCCL runs it on its own stream).  ``wait()`` orders the caller's
    current stream after it and returns (mel [world*Bl,T,M], mel_len [world*Bl]).  The packed send / receive buffers
    are owned by this object, so the producer may overwrite its ``mel`` workspace as soon as the call returns."""

    def __init__(self, work, out, T, M, keep):
        self._work, self._out, self._T, self._M, self._keep = work, out, T, M, keep

    def wait(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        return unpack_mels(self._out, self._T, self._M)


def allgather_mels_async(mel: torch.Tensor, mel_len: torch.Tensor, group=None, force: bool = False) -> PendingGather:
    """``allgather_mels`` issued asynchronously: the collective of batch i overlaps the text-side kernels of batch
    i+1 (the xGMI links are idle otherwise).  Callers ``wait()`` before they launch the next persistent denoiser
    stack — that kernel needs every CU, so it must not share the GPU with RCCL's kernels — and before they read the
    result.  ``force`` runs the collective on a 1-rank group too (single-GPU test of the RCCL call sequence)."""
    Bl, T, M = mel.shape
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return PendingGather(None, pack_mels(mel, mel_len), T, M, None)
    world = dist.get_world_size(group)
    buf = pack_mels(mel, mel_len)
    out = torch.empty(world * Bl, buf.shape[1], dtype=buf.dtype, device=buf.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.chunk(world, 0))
        work = dist.all_gather(parts, buf, group=group, async_op=True)
        return PendingGather(work, out, T, M, (buf, parts))
    work = dist.all_gather_into_tensor(out, buf, group=group, async_op=True)
    return PendingGather(work, out, T, M, buf)


def pack_pcm(pcm: torch.Tensor, wav_len: torch.Tensor) -> torch.Tensor:
    """int16 [Bl,N] + int64 [Bl] -> one int16 buffer [Bl, N4 + 4] (N4 = N rounded up to a multiple of 4): the sample count
    rides behind each row as an int64 in four int16 slots — the layout of cmtts_allgather_pcm (csrc/rccl_gather.hip)."""
    Bl, N = pcm.shape
    n4 = (N + 3) // 4 * 4
    buf = torch.zeros(Bl, n4 + 4, dtype=torch.int16, device=pcm.device)
    buf[:, :N] = pcm
    buf[:, n4:].view(torch.int64)[:, 0] = wav_len.to(torch.int64)
    return buf


def unpack_pcm(buf: torch.Tensor, N: int) -> Tuple[torch.Tensor, torch.Tensor]:
    n4 = buf.shape[1] - 4
    return buf[:, :N], buf[:, n4:].contiguous().view(torch.int64)[:, 0]


def allgather_pcm(pcm: torch.Tensor, wav_len: torch.Tensor, group=None, force: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """End-to-end wav jobs (BASELINE.json configs[4]; SURVEY.md §8e: "cfg5 gathers int16 wav [16, 1024*256] = 8.4 MB/rank
    instead"): what is collated is vocoder_infer's output (utils/model.py:187-205) — every rank contributes its padded int16
    block pcm [Bl,N] and the valid sample counts wav_len [Bl] (= mel_len * hop) and receives [world*Bl, N] + [world*Bl] in
    rank order: ONE all-gather of bytes (neither RCCL nor gloo moves int16 natively).  `force` runs the collective on a
    1-rank group too (single-GPU check of the RCCL call sequence)."""
    assert pcm.dtype == torch.int16 and pcm.dim() == 2
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return pcm, wav_len.to(torch.int64)
    world = dist.get_world_size(group)
    Bl, N = pcm.shape
    buf = pack_pcm(pcm, wav_len)
    send = buf.view(torch.uint8).reshape(-1)
    out = torch.empty(world, send.numel(), dtype=torch.uint8, device=buf.device)
    if dist.get_backend(group) == "gloo":
        dist.all_gather(list(out.unbind(0)), send, group=group)
    else:
        dist.all_gather_into_tensor(out.reshape(-1), send, group=group)
    return unpack_pcm(out.view(torch.int16).reshape(world * Bl, buf.shape[1]), N)


def restore_pcm_order(pcm: torch.Tensor, wav_len: torch.Tensor, deal: Sequence[Sequence[int]], n_items: int):
    """Gathered rows (rank-major, rank r's rows in the order of deal[r]; -1 = filler) -> list of per-utterance int16 tensors
    trimmed to their sample counts, in the ORIGINAL utterance order (the list vocoder_infer returns, utils/model.py:199-205)."""
    out = [None] * n_items
    flat = [i for r in deal for i in r]
    assert pcm.shape[0] == len(flat), (pcm.shape, len(flat))
    for row, i in enumerate(flat):
        if i >= 0:
            out[i] = pcm[row, : int(wav_len[row])]
    assert all(o is not None for o in out)
    return out


def allgather_buckets(mels, group=None, force: bool = False):
    """configs[3]: every bucket of a ragged shard in ONE all-gather.  mels: {bucket: (mel [n_b, bucket, M], mel_len
    [n_b])} with the same n_b on every rank (plan_shards guarantees it).  The blocks are packed back to back (each as
    pack_mels lays it out: mel_len rides in the same buffer) into one flat fp32 buffer per rank, gathered once, and split
    again.  Returns {bucket: (mel [world*n_b, bucket, M], mel_len [world*n_b])} in rank order — what restore_order takes.
    `force` runs the collective on a 1-rank group too (single-GPU check of the RCCL call sequence)."""
    buckets = sorted(mels)
    shapes = {b: tuple(mels[b][0].shape) for b in buckets}
    parts = [pack_mels(mels[b][0], mels[b][1]).reshape(-1) for b in buckets]
    sizes = [p.numel() for p in parts]
    buf = torch.cat(parts) if len(parts) > 1 else parts[0]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized() or (world == 1 and not force):
        out = buf[None]
    else:
        out = torch.empty(world, buf.numel(), dtype=buf.dtype, device=buf.device)
        if dist.get_backend(group) == "gloo":
            dist.all_gather(list(out.unbind(0)), buf, group=group)
        else:
            dist.all_gather_into_tensor(out.reshape(-1), buf, group=group)
    res, off = {}, 0
    for b, n in zip(buckets, sizes):
        nb, T, M = shapes[b]
        blk = out[:, off:off + n].reshape(out.shape[0] * nb, T * M + 1)
        res[b] = unpack_mels(blk, T, M)
        off += n
    return res
