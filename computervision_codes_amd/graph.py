"""hipGraph capture of a fixed-shape forward (launch-bound inner loops: the 41-layer TCN is ~100 launches of a
few microseconds each).  All libmt4hip entry points only enqueue on the caller's stream and never allocate, so a
whole model forward captures; torch provides the capture stream and the graph-private memory pool (plumbing)."""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedForward:
    """Capture `fn(*static_inputs)` once; `__call__(*inputs)` copies the inputs into the static buffers and replays.
    Outputs are the static output tensors of the captured run (overwritten by the next replay)."""

    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 2):
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up on a side stream (allocator + lazy module loading)
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs: torch.Tensor):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out


class SegmentedGraph:
    """`fn(cut, *static_inputs)` captured as CONSECUTIVE hipGraphs that share one memory pool, cut wherever `fn` calls `cut(tag)`; a replay runs
    the segments in order and calls `on_cut(tag)` between them.  For data-parallel training under graph replay: the backward is cut where a
    gradient bucket is complete, the bucket's all-reduce (RCCL, its own stream) is issued between two replays and runs beside the next
    segment -- one captured graph could only be followed by ONE flat all-reduce behind the whole backward."""

    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 2):
        self.static_in = [t.clone() for t in example_inputs]
        self.segments = []                      # [(CUDAGraph, tag of the cut behind it or None)]
        self._capturing = False
        self._cur = None
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):   # warm-up (allocator, lazy module loading); cuts are no-ops
            for _ in range(warmup):
                fn(self.cut, *self.static_in)
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()
        self._pool = torch.cuda.graph_pool_handle()
        with torch.cuda.stream(self._stream):
            self._begin()
            self._capturing = True
            try:
                self.static_out = fn(self.cut, *self.static_in)
            finally:
                self._capturing = False
            self._cur.capture_end()
            self.segments.append((self._cur, None))
            self._cur = None
        torch.cuda.current_stream().wait_stream(self._stream)
        torch.cuda.synchronize()

    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self._pool)

    def cut(self, tag) -> None:
        """called by `fn`: everything enqueued so far forms a segment; `tag` is handed to `on_cut` behind its replay"""
        if not self._capturing:
            return
        self._cur.capture_end()
        self.segments.append((self._cur, tag))
        self._begin()

    def __call__(self, *inputs: torch.Tensor, on_cut: Callable = None):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        for g, tag in self.segments:
            g.replay()
            if tag is not None and on_cut is not None:
                on_cut(tag)
        return self.static_out
