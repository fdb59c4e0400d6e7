"""Multi-GPU data path: one process per GPU, chunks sharded contiguously, one all-gather of records.

The reference processes chunks concurrently and order-independently (4 threads of network calls,
ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:623-641) and re-orders only at merge
(:647-654).  Chunks share no state, so here every rank computes the same integer plan, encodes
only its contiguous shard on its own GPU, and the fixed-size per-chunk records are exchanged with a
single `all_gather` (RCCL over xGMI under backend "nccl"; gloo in CPU tests).  There is no other
collective on the data path.
"""

from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.distributed as dist

RECORD_WORDS = 5  # chunk_idx, valid, start_sample, end_sample, n_frames
# A record may carry the chunk's CTC tokens after its header: word RECORD_WORDS holds the token count
# (-1: the engine has no head), the following words hold (frame << 32 | token id), one per token.


def shard_bounds(n_chunks: int, world: int, rank: int) -> tuple[int, int]:
    from omnilingual_asr import _hip
    return _hip.shard_range(n_chunks, world, rank)


def max_shard(n_chunks: int, world: int) -> int:
    return -(-n_chunks // world)


def pack_records(records, capacity: int, token_capacity: int = 0) -> torch.Tensor:
    """[capacity, RECORD_WORDS (+ 1 + token_capacity)] int64; unused rows have valid = 0."""
    width = RECORD_WORDS + (1 + token_capacity if token_capacity > 0 else 0)
    buf = np.zeros((capacity, width), dtype=np.int64)
    for i, r in enumerate(records):
        buf[i, 0] = r.chunk_idx
        buf[i, 1] = 1
        buf[i, 2] = r.start_sample
        buf[i, 3] = r.end_sample
        buf[i, 4] = r.n_frames
        if token_capacity > 0:
            if getattr(r, "tokens", None) is None:
                buf[i, RECORD_WORDS] = -1
            else:
                n = len(r.tokens)
                if n > token_capacity:
                    raise ValueError("record has %d tokens, capacity %d" % (n, token_capacity))
                buf[i, RECORD_WORDS] = n
                buf[i, RECORD_WORDS + 1: RECORD_WORDS + 1 + n] = \
                    (np.asarray(r.frames, dtype=np.int64) << 32) | np.asarray(r.tokens, dtype=np.int64)
    return torch.from_numpy(buf)


def unpack_records(gathered: torch.Tensor):
    """Rows with valid = 1 from a [world * capacity, width] tensor, ordered by chunk index — the
    order the reference's merge produces, since chunk starts grow with the index."""
    from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord

    arr = gathered.cpu().numpy()
    arr = arr.reshape(-1, arr.shape[-1])
    rows = arr[arr[:, 1] == 1]
    rows = rows[np.argsort(rows[:, 0], kind="stable")]
    out = []
    for r in rows:
        tokens = frames = None
        if arr.shape[1] > RECORD_WORDS and r[RECORD_WORDS] >= 0:
            n = int(r[RECORD_WORDS])
            packed = r[RECORD_WORDS + 1: RECORD_WORDS + 1 + n]
            tokens = (packed & 0xFFFFFFFF).astype(np.int32)
            frames = (packed >> 32).astype(np.int32)
        out.append(ChunkRecord(int(r[0]), int(r[2]), int(r[3]), int(r[4]), tokens, frames))
    return out


def all_gather_records(local: torch.Tensor, group=None) -> torch.Tensor:
    """The one collective of the path: equal-size record blocks from every rank."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if backend == "nccl":
        local = local.cuda()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat(parts, dim=0)


def encode_sharded(pipeline, pcm: np.ndarray, bounds: np.ndarray, group=None) -> List:
    """Encode this rank's shard of `bounds` with `pipeline` and return the records of all ranks."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(len(bounds), world, rank)
    mine = pipeline._encode_chunks(pcm, bounds[lo:hi], first_idx=lo)
    return exchange_records(mine, bounds, group)


def exchange_records(mine, bounds: np.ndarray, group=None) -> List:
    """This rank's records (of its shard of `bounds`) in, the records of all ranks out, through the one all-gather."""
    world = dist.get_world_size(group)
    # every rank derives the same token capacity from the plan: one token per 20 ms frame at most
    longest = int(max(e - s for s, e in bounds)) if len(bounds) else 0
    token_capacity = max(1, longest // 320)
    packed = pack_records(mine, max_shard(len(bounds), world), token_capacity)
    return unpack_records(all_gather_records(packed, group))
