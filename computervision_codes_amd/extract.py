"""Extraction harness: batches frames in file order, no shuffle, drop_last False, one
feature block [N,D] per video (`Spatial_cnn/test.py:143-177`), sharded over ranks by whole videos.

Videos are independent units, so N ranks need no data-path collective: every rank runs its own videos and
rank 0 merges the per-rank dicts once at the end (`gather_feats`, an object gather on the host) to write the
single pickle the temporal stage reads.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, List, Mapping, Sequence

import numpy as np
import torch


def shard_videos(video_keys: Sequence[str], n_frames: Sequence[int], rank: int, world: int) -> List[int]:
    """Greedy longest-first balancing of whole videos over ranks; deterministic on every rank.
    Returns the indices (into video_keys) owned by `rank`, in original order."""
    if len(video_keys) != len(n_frames):
        raise ValueError("video_keys / n_frames length mismatch")
    order = sorted(range(len(video_keys)), key=lambda i: (-int(n_frames[i]), i))
    load = [0] * world
    owner = [0] * len(video_keys)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(n_frames[i])
    return [i for i in range(len(video_keys)) if owner[i] == rank]


def extract_video(frames: torch.Tensor, forward: Callable[[torch.Tensor], torch.Tensor], batch: int) -> np.ndarray:
    """frames [N,...] in file order -> float32 [N,D]; `forward` maps a batch to its [b,D] features."""
    out = []
    for s in range(0, frames.shape[0], batch):   # last batch short: drop_last=False
        out.append(forward(frames[s:s + batch]).detach().float().cpu())
    return torch.vstack(out).numpy() if out else np.zeros((0, 0), np.float32)


_POOLS: Dict[tuple, ThreadPoolExecutor] = {}        # helper threads by (device, prefetch depth), kept for the life of the process: a thread's pinned
                                                     # staging buffer (`pngdec._staging`, >= 64 MB per thread) is page-locked once, not once per video


_SIDE_STREAMS: Dict[tuple, list] = {}


def _shutdown_pools():
    for pool in _POOLS.values():
        pool.shutdown(wait=False, cancel_futures=True)
    _POOLS.clear()


import atexit  # noqa: E402

atexit.register(_shutdown_pools)


def iter_chunks(spans: Sequence[tuple], load_chunk: Callable[[int, int], torch.Tensor], prefetch=1):
    """Yield `load_chunk(s, e)` for every span in order.  With `prefetch` = k > 0 the next k spans are read and decoded on helper threads
    while the caller works on this one (the reference's DataLoader workers run ahead of the model the same way, `Spatial_cnn/test.py:240-241`);
    k = 2 lets the host part of one load (file reads, gathering the compressed bytes) overlap the device part of the load before it.
    Every load in flight launches on a SIDE stream of its own (a loader that ends in a blocking status read -- the device PNG decoder -- then
    waits for its own kernels only, not for every model pass the caller has queued, and the device parts of the k loads -- upload, inflate,
    unfilter, resize -- co-run: the inflate is latency-bound per frame, 75 ms per call however many frames ride along).  A chunk is handed over
    with an event the CALLER'S CURRENT stream waits on (and `record_stream` for that stream): a consumer that fans out over streams of its own
    (`extract_u8(streams > 1)`) must order them behind the current stream, as `VideoNas.extract_u8` does (`st.wait_stream(main)`).
    `spans` are argument tuples of `load_chunk`."""
    depth = int(prefetch)
    if depth <= 0 or len(spans) < 2 or not torch.cuda.is_available():
        for sp in spans:
            yield load_chunk(*sp)
        return
    dev, cur = torch.cuda.current_device(), torch.cuda.current_stream()
    depth = min(depth, len(spans))
    # one side stream per load in flight: their device parts (upload, inflate, resize) co-run.  Kept for the life of the process like the helper
    # threads: torch's caching allocator pools memory per stream, so fresh streams per call meant a hipMalloc for every buffer of every load
    sides = _SIDE_STREAMS.get((dev, depth))
    if sides is None:
        sides = _SIDE_STREAMS[(dev, depth)] = [torch.cuda.Stream() for _ in range(depth)]

    def ahead(i):
        torch.cuda.set_device(dev)
        st = sides[i % depth]
        with torch.cuda.stream(st):
            fr = load_chunk(*spans[i])
            ev = torch.cuda.Event()
            ev.record(st)
        return fr, ev
    pool = _POOLS.get((dev, depth))
    if pool is None:
        pool = _POOLS[(dev, depth)] = ThreadPoolExecutor(depth, thread_name_prefix=f"mt4-load-{dev}")
    pending = [pool.submit(ahead, i) for i in range(depth)]
    try:
        for i in range(len(spans)):
            fr, ev = pending.pop(0).result()
            if i + depth < len(spans):
                pending.append(pool.submit(ahead, i + depth))
            cur.wait_event(ev)
            if torch.is_tensor(fr) and fr.is_cuda:
                fr.record_stream(cur)           # allocated on a side stream, consumed on the caller's
            yield fr
    finally:
        for f in pending:                       # (the consumer stopped early: let the loads in flight finish before their buffers go)
            f.cancel()
        for f in pending:
            if not f.cancelled():
                try:
                    f.result()
                except Exception:
                    pass


def extract_videos_device(model, videos, device_batch: int = 512, streams: int = 1, prefetch=1, load_batch: int = None):
    """Several videos through the spatial extractor with ONE loader pipeline across them (`Spatial_cnn/test.py:266-268` loops over the videos;
    its DataLoader workers start each video cold): `videos` = sequence of (key, n_frames, load_chunk); the spans of all videos form one
    sequence for `iter_chunks`, so while video k's last passes run the first loads of video k + 1 are already being read and decoded -- the
    pipeline fills once per run, not once per video.  Yields (key, feat [N,D] float32 ndarray, logits (i, v, t, ivt) float32 ndarrays) per
    video, in order; features and logits of a video stay on the device until it ends, then cross to the host ONCE through pinned memory."""
    load_batch = max(device_batch, load_batch or device_batch) // device_batch * device_batch      # whole passes per load
    spans = [(vi, s, min(n, s + load_batch)) for vi, (_, n, _) in enumerate(videos) for s in range(0, n, load_batch)]
    last = {vi: e for vi, _, e in spans}

    def finish(feats, logits):
        if not feats:
            return np.zeros((0, 0), np.float32), tuple(np.zeros((0, 0), np.float32) for _ in range(4))
        dev_out = [torch.cat(feats).float()] + [torch.cat(l).float() for l in logits]
        host = [torch.empty(t.shape, dtype=torch.float32, pin_memory=True) for t in dev_out]
        for h, d in zip(host, dev_out):
            h.copy_(d, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return host[0].numpy(), tuple(h.numpy() for h in host[1:])
    done = 0
    feats, logits = [], [[], [], [], []]
    chunks = iter_chunks(spans, lambda vi, s, e: videos[vi][2](s, e), prefetch)
    for (vi, s, e), span in zip(spans, chunks):
        while done < vi:                                   # (videos without frames in front of this one)
            yield (videos[done][0],) + finish([], None)
            done += 1
        for p0 in range(0, span.shape[0], device_batch):
            (_, li), (_, lv), (_, lt), (feat, livt) = model.extract_u8(span[p0:p0 + device_batch], streams=streams)
            feats.append(feat)
            for acc, lg in zip(logits, (li, lv, lt, livt)):
                acc.append(lg)
        if e == last[vi]:
            yield (videos[vi][0],) + finish(feats, logits)
            feats, logits = [], [[], [], [], []]
            done = vi + 1
    while done < len(videos):
        yield (videos[done][0],) + finish([], None)
        done += 1


def extract_video_device(model, n_frames: int, load_chunk: Callable[[int, int], torch.Tensor], device_batch: int = 512, streams: int = 1,
                         prefetch=1, load_batch: int = None):
    """One video through the spatial extractor the MI355X way (`Spatial_cnn/test.py:143-177` restated): frames [s, e) arrive as uint8
    device tensors from `load_chunk(s, e)` in file order, `device_batch` of them per pass (a frame's result does not depend on the batch
    it rides in -- bit-exact, tests/test_gpu_models.py -- so the reference's `--batch` need not bound the launch size); features and
    the four heads' logits stay on the device until the video ends, then cross to the host ONCE through pinned memory.  `load_batch`
    (a multiple of `device_batch`): frames per `load_chunk` call when the loader wants more than one pass at a time (the device PNG decoder
    runs one wave per frame and fills the GPU from ~2500 frames on).
    Returns (feat [N,D] float32 ndarray, logits (i, v, t, ivt) float32 ndarrays)."""
    for _, feat, lgs in extract_videos_device(model, [(None, n_frames, load_chunk)], device_batch, streams, prefetch, load_batch):
        return feat, lgs


def gather_feats(local: Mapping[str, np.ndarray], group=None) -> Dict[str, np.ndarray]:
    """Merge per-rank {video -> [N,D]} dicts on every rank (host-side object gather; no-op for 1 rank)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local)
    parts: List[Dict[str, np.ndarray]] = [None] * dist.get_world_size(group)  # type: ignore
    dist.all_gather_object(parts, dict(local), group=group)
    merged: Dict[str, np.ndarray] = {}
    for p in parts:
        for k, v in p.items():
            if k in merged:
                raise ValueError(f"video {k} extracted by two ranks")
            merged[k] = v
    return merged


def extract_dataset(videos: Mapping[str, torch.Tensor], forward: Callable[[torch.Tensor], torch.Tensor], batch: int,
                    rank: int = 0, world: int = 1, group=None) -> Dict[str, np.ndarray]:
    """videos: {key -> frames [N,...]} (same mapping on every rank).  Returns the merged {key -> [N,D]} with
    keys in the original order."""
    keys = list(videos.keys())
    mine = shard_videos(keys, [videos[k].shape[0] for k in keys], rank, world)
    local = {keys[i]: extract_video(videos[keys[i]], forward, batch) for i in mine}
    merged = gather_feats(local, group)
    return {k: merged[k] for k in keys}
