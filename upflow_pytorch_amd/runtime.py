"""HIP-graph runtime for fixed-shape inference.

One UPFlow forward is a few hundred small launches (MIOpen convs, concatenations, the hand-written
operators).  At the coarse pyramid levels every one of them is launch-latency bound, so the whole
forward is captured ONCE into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed: no python, no
per-launch host work, no allocator traffic in the steady state.  All libupflow_hip.so entry points
only enqueue work on the given stream, which is what makes them capturable.
"""
import torch


class GraphedInference:
    """net(input_dict) for a fixed (B,H,W): static input buffers, captured forward, static outputs.

        runner = GraphedInference(net, B, H, W)
        out = runner(im1, im2)          # dict of tensors owned by the runner (valid until next call)
    """

    def __init__(self, net, B, H, W, in_dtype=torch.float32, device=None, warmup=3, check_weights=True):
        self.net = net
        self.check_weights = check_weights
        p = next(net.parameters())
        self.device = device if device is not None else p.device
        self.im1 = torch.zeros(B, 3, H, W, dtype=in_dtype, device=self.device)
        self.im2 = torch.zeros_like(self.im1)
        self.graph = None
        self.out = None
        self._capture(warmup)

    def _forward(self):
        return self.net({'im1': self.im1, 'im2': self.im2, 'if_loss': False})

    def _capture(self, warmup):
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # MIOpen chooses its solvers and workspaces before capture
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # a live process group (bench.py --gpus N, DDP serving) has a watchdog thread that polls its events while this thread
        # captures: under the default (global) capture mode that hipEventQuery aborts the capture (train.Trainer does the same)
        import torch.distributed as dist
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'
        from .parallel import no_gc_during_capture
        with no_gc_during_capture(), torch.no_grad(), torch.cuda.graph(g, capture_error_mode=mode):
            self.out = self._forward()
        self.graph = g
        # The graph has the addresses of the PACKED weight operands baked in (allocated by the warm-up forwards, outside the graph's
        # pool).  They are re-packed — and the old tensors freed — when a parameter changes (load_state_dict, an optimizer step):
        # hold them, so that a replay after that reads stale weights rather than recycled memory, and refuse such a replay.
        from .model.pwc_modules import packed_operands
        self._keepalive = packed_operands(self.net)
        self._params = list(self.net.parameters())
        self._weights_key = self._weights_snapshot()

    def _weights_snapshot(self):             # (~10 us for the 80 parameters)
        ps = self._params
        return (tuple(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr())

    def recapture(self, warmup=1):
        """Capture again (after the weights changed); the static inputs keep their contents, `out` is a new dict."""
        self.graph = None
        self._capture(warmup)

    def load(self, im1, im2):
        self.im1.copy_(im1, non_blocking=True)
        self.im2.copy_(im2, non_blocking=True)

    def replay(self):
        if self.check_weights and self._weights_snapshot() != self._weights_key:
            raise RuntimeError('GraphedInference: the network\'s parameters changed after the capture (the graph reads the packed copies '
                               'made then) — call recapture()')
        self.graph.replay()
        return self.out

    def __call__(self, im1, im2):
        self.load(im1, im2)
        return self.replay()


class PipelinedInference:
    """Throughput mode: `streams` captured forwards of the SAME network (one GraphedInference each, with its own static
    input / output buffers and its own graph memory pool; the packed weights are shared, read-only) replayed round-robin on
    as many HIP streams, so that several steps are in flight at once.

    Why: one UPFlow step is a chain of ~170 dependent launches, and at the three coarse pyramid levels (6x20 .. 24x80 pixels)
    each of them occupies a few dozen of the 256 CUs for ~10 us — a fifth of the step's time on a tenth of the chip, which no
    single-step schedule can fill (the levels depend on each other).  A second, independent step can: its fine-level
    convolutions run under the other step's coarse levels.  Measured on MI355X, config 2 (384x1280, bf16, batch 4 per step):
    3.14 ms per step with one step in flight, 2.50-2.57 ms with two (tools/stream_probe.py) — the per-step LATENCY grows
    (~5 ms), the throughput by 22-25 %.  Every step is the full forward on its own batch; nothing is shared between steps
    but the weights.

        pipe = PipelinedInference(net, B, H, W, streams=2)
        t = pipe.submit(im1, im2)        # enqueue on the next stream (returns a ticket); inputs are copied on that stream
        out = pipe.result(t)             # waits for that step only; tensors are owned by the pipe until its slot is reused
        for out in pipe.map(loader):     # or: a whole iterable of (im1, im2) with `streams` steps in flight, outputs in order
    """

    def __init__(self, net, B, H, W, streams=2, in_dtype=torch.float32, device=None, warmup=3, check_weights=True):
        p = next(net.parameters())
        self.device = device if device is not None else p.device
        self.n = int(streams)
        if self.n < 1:
            raise ValueError('streams must be >= 1')
        self.runners = [GraphedInference(net, B, H, W, in_dtype=in_dtype, device=self.device, warmup=warmup if i == 0 else 1, check_weights=check_weights)
                        for i in range(self.n)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self.events = [None] * self.n
        self._next = 0
        torch.cuda.synchronize(self.device)

    def load(self, slot, im1, im2):
        """Copy a batch into slot's static inputs on the slot's stream (ordered before its next replay).  The copy waits for
        the caller's current stream — the producer of im1 / im2, e.g. a host-to-device transfer that is still in flight — and
        the sources are recorded on the slot's stream, so that a temporary (`x.to(device)`) is not recycled by the allocator
        before the copy has read it."""
        st = self.streams[slot]
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            self.runners[slot].load(im1, im2)
        for t in (im1, im2):
            if t.is_cuda:
                t.record_stream(st)

    def replay(self, slot=None):
        """Replay one step on the next (or the given) slot without touching its inputs; returns the slot."""
        if slot is None:
            slot = self._next
            self._next = (self._next + 1) % self.n
        st = self.streams[slot]
        # whoever consumed the slot's previous outputs (or wrote its inputs) did so on the caller's current stream: the new replay
        # overwrites those buffers, so it is ordered behind that work — as load() does for the inputs
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            self.runners[slot].replay()
            ev = torch.cuda.Event()
            ev.record(self.streams[slot])
        self.events[slot] = ev
        return slot

    def submit(self, im1, im2):
        slot = self._next
        self.load(slot, im1, im2)
        return self.replay()

    def result(self, slot):
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        return self.runners[slot].out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def map(self, batches):
        """`for out in pipe.map(pairs)`: runs every (im1, im2) of the iterable with `streams` steps in flight and yields the output
        dicts IN ORDER.  A yielded dict is the slot's STATIC output: advancing the generator re-submits into that very slot (the
        oldest pending step's slot is the next one to be reused), so a yielded dict is valid only until the generator is advanced
        ONCE — consume it, or clone what you keep, before asking for the next item (`list(pipe.map(...))` would hold overwritten
        tensors)."""
        from collections import deque
        pending = deque()
        for im1, im2 in batches:
            if len(pending) == self.n:
                yield self.result(pending.popleft())      # (the slot this submit is about to reuse)
            pending.append(self.submit(im1, im2))
        while pending:
            yield self.result(pending.popleft())

    def recapture(self, warmup=1):
        """Capture every slot again (after the weights changed): waits for the steps in flight first."""
        self.synchronize()
        for r in self.runners:
            r.recapture(warmup)
        self.events = [None] * self.n
        torch.cuda.synchronize(self.device)


class ShapeCachedInference:
    """net(input_dict) for the reference's EVALUATION workload: batch 1 (or any batch), frames whose size changes from
    sequence to sequence (/root/reference/test.py:40-47: "eval batch size should be 1 ... image size may be different for
    different sequence"; KITTI 2012 / 2015 ship 375x1242, 370x1224, 374x1238, 376x1241 ...).  One GraphedInference per
    (B, H, W, input dtype), captured the first time the shape is seen and replayed afterwards: an eager forward of one
    375x1242 pair is ~170 ctypes launches = 2.9 ms of host time for 1.6 ms of GPU work; the replay is one call.

        runner = ShapeCachedInference(net)                 # net: UPFlow_net in eval mode, on the GPU
        out = runner(im1, im2)                             # dict of tensors owned by the runner: valid until the next call
                                                           # with the same shape (clone what you keep)
    `max_shapes` bounds the cache (least recently used graph dropped: a captured step holds ~0.4 GB of buffers at 375x1242);
    a change of the weights (load_model, an optimizer step) drops every graph (GraphedInference.check_weights semantics, but
    transparently: the next call re-captures)."""

    def __init__(self, net, max_shapes=8, warmup=2):
        from collections import OrderedDict
        self.net = net
        self.max_shapes = int(max_shapes)
        self.warmup = int(warmup)
        self._runners = OrderedDict()
        self.captures = 0

    def _key(self, im1):
        return (tuple(im1.shape), im1.dtype, im1.device)

    def __call__(self, im1, im2):
        if im1.shape != im2.shape or im1.dim() != 4:
            raise ValueError('ShapeCachedInference: two [B,3,H,W] frames of one size expected, got %s / %s' % (tuple(im1.shape), tuple(im2.shape)))
        key = self._key(im1)
        r = self._runners.get(key)
        if r is not None and r._weights_snapshot() != r._weights_key:
            self._runners.clear()                      # the weights changed: every graph reads packed copies of the old ones
            r = None
        if r is None:
            if len(self._runners) >= self.max_shapes:
                self._runners.popitem(last=False)
            B, _, H, W = im1.shape
            r = GraphedInference(self.net, B, H, W, in_dtype=im1.dtype, device=im1.device, warmup=self.warmup)
            self._runners[key] = r
            self.captures += 1
        else:
            self._runners.move_to_end(key)
        return r(im1, im2)

    def shapes(self):
        return [k[0] for k in self._runners]


class PipelinedEvaluation:
    """The evaluation loop with several frame pairs in flight.  The reference evaluates one pair at a time (test.py:40-47: forward,
    then the metrics on the host); a single 375x1242 pair leaves most of the chip idle at its coarse pyramid levels (1.65 ms per pair
    through a replayed graph).  Here up to `streams` pairs — of possibly DIFFERENT sizes — are in flight on as many HIP streams, each
    on a captured graph of its frame size (a per-size pool of `streams` GraphedInference slots, captured on first sight), and results
    come back in submission order:

        ev = PipelinedEvaluation(net, streams=4)
        for out in ev.map(pairs):            # pairs: iterable of (im1, im2); out: dict of tensors owned by the slot — valid until
            ...                              # the generator is advanced again (clone what you keep longer)

    375x1242, batch 1, bf16: ~0.95 ms per pair with four in flight against 1.65 ms one at a time and 3.0 ms eager.  Every pair's
    result is bit-identical to the same pair run alone (tests/test_hip_net.py)."""

    def __init__(self, net, streams=4, max_shapes=4, warmup=2):
        from collections import OrderedDict
        self.net = net
        self.n = int(streams)
        self.max_shapes = int(max_shapes)
        self.warmup = int(warmup)
        self.device = next(net.parameters()).device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self._pools = OrderedDict()            # (shape, dtype) -> [GraphedInference] * n   (slot i always runs on stream i)
        self._busy = [None] * self.n           # per stream: the event of the step in flight on it
        self._next = 0
        self.captures = 0

    def _pool(self, im1):
        key = (tuple(im1.shape), im1.dtype)
        pool = self._pools.get(key)
        if pool is not None and pool[0]._weights_snapshot() != pool[0]._weights_key:
            self.synchronize()
            self._pools.clear()
            pool = None
        if pool is None:
            self.synchronize()                 # (captures use the legacy stream semantics of the warm-up: nothing else in flight)
            if len(self._pools) >= self.max_shapes:
                self._pools.popitem(last=False)
            B, _, H, W = im1.shape
            pool = [GraphedInference(self.net, B, H, W, in_dtype=im1.dtype, device=self.device, warmup=self.warmup if i == 0 else 1)
                    for i in range(self.n)]
            self._pools[key] = pool
            self.captures += 1
        else:
            self._pools.move_to_end(key)
        return pool

    def submit(self, im1, im2):
        """Enqueue one pair on the next stream; returns a ticket for result()."""
        if im1.shape != im2.shape or im1.dim() != 4:
            raise ValueError('PipelinedEvaluation: two [B,3,H,W] frames of one size expected')
        pool = self._pool(im1)
        i = self._next
        self._next = (self._next + 1) % self.n
        st, r = self.streams[i], pool[i]
        st.wait_stream(torch.cuda.current_stream(self.device))        # the producer of im1 / im2, the consumer of the slot's last outputs
        with torch.cuda.stream(st):
            r.load(im1, im2)
            r.replay()
            ev = torch.cuda.Event()
            ev.record(st)
        for t in (im1, im2):
            if t.is_cuda:
                t.record_stream(st)
        self._busy[i] = ev
        return (r, ev)

    def result(self, ticket):
        r, ev = ticket
        ev.synchronize()
        return r.out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def map(self, pairs):
        from collections import deque
        pending = deque()
        for im1, im2 in pairs:
            if len(pending) == self.n:
                yield self.result(pending.popleft())
            pending.append(self.submit(im1, im2))
        while pending:
            yield self.result(pending.popleft())
