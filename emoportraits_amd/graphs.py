"""hipGraph capture of launch-bound kernel sequences.

A driver frame is ~150 kernel launches in the hot path plus ~130 in the two ResNet-18 embedders; at batch 1 the GPU
finishes most of them faster than the host can enqueue the next one (ctypes call + output allocation per launch).  The
sequences are static -- same kernels, same shapes, same buffers for a given batch size -- so they are captured once per
input signature into a hipGraph and replayed with one launch.  Capture goes through torch's stream-capture plumbing
(`torch.cuda.CUDAGraph` = hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch on ROCm): the C-ABI kernels are
enqueued on `torch.cuda.current_stream()`, which is the capturing stream inside the context, and intermediate buffers
come from the graph's private pool of the caching allocator.

Rules for a capturable callable: device tensors in, device tensors out, no host<->device copies and no host
synchronisation inside (constants must be resident before capture: `warmup` eager calls take care of lazy packing).
"""
import torch


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return [out], lambda ts: ts[0]
    if isinstance(out, (tuple, list)):
        n = len(out)
        kind = type(out)
        return list(out), lambda ts: kind(ts[:n])
    raise TypeError("a graphed callable must return a tensor or a tuple/list of tensors")


class Graphed:
    """fn(*device_tensors) -> tensor(s), replayed from a hipGraph keyed by the input shapes.

    Outputs live in the graph's static buffers: by default they are cloned so they survive the next replay; pass
    clone_outputs=False to hand out the static buffers themselves (valid until the next call with the same signature)."""

    def __init__(self, fn, warmup=2, clone_outputs=True, max_signatures=8, eager_calls=0):
        """eager_calls: calls per input signature that run fn directly before the signature is captured -- a wrapper that
        turns graphs on by default pays the capture (two warm-up runs + one captured run + static buffers) only for shapes
        that actually recur"""
        self.fn, self.warmup, self.clone_outputs, self.max_signatures = fn, warmup, clone_outputs, max_signatures
        self.eager_calls = eager_calls
        self._cache = {}
        self._seen = {}

    def _capture(self, tensors):
        static_in = [t.clone() for t in tensors]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up off the default stream (lazy weight packing etc.)
            for _ in range(self.warmup):
                self.fn(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: only this thread's calls are checked against the capture -- other threads of the process (the RCCL
        # watchdog of a multi-GPU run, the pinned-memory copier of animate_frames) may touch the runtime meanwhile
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = self.fn(*static_in)
        outs, rebuild = _flatten(out)
        return static_in, graph, outs, rebuild

    def __call__(self, *tensors):
        key = tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors)
        entry = self._cache.get(key)
        if entry is None and self._seen.get(key, 0) < self.eager_calls:
            if len(self._seen) > 64:
                self._seen.clear()
            self._seen[key] = self._seen.get(key, 0) + 1
            return self.fn(*tensors)
        if entry is None:
            if len(self._cache) >= self.max_signatures:
                self._cache.pop(next(iter(self._cache)))
            entry = self._cache[key] = self._capture(tensors)
        static_in, graph, outs, rebuild = entry
        for s, t in zip(static_in, tensors):
            s.copy_(t, non_blocking=True)
        graph.replay()
        return rebuild([o.clone() for o in outs] if self.clone_outputs else outs)

    def signatures(self):
        return list(self._cache)
