"""Unsupervised training step (BASELINE config 3) — one process per GPU, DDP over RCCL/xGMI.

What the reference intends (its published scripts do not run: SURVEY.md §3.3):
    optimiser  Adam(lr=1e-4, amsgrad=True, weight_decay=1e-4) + ExponentialLR(gamma)
                                                            scripts/simple_train.py:121-122
    batch      {'im1','im2' (crops), 'im1_raw','im2_raw' (un-cropped), 'start', 'if_loss': True}
                                                            scripts/ex_runner.py:146-147
    loss       photo_loss.mean() + smooth_loss.mean() [+ census_loss.mean()] [+ msd_loss.mean()]
                                                            scripts/ex_runner.py:151-160, simple_train.py:23-54
Multi-GPU: the reference wraps the net in single-process nn.DataParallel (utils/tools.py:140); here the
global batch is sharded across ranks, each rank runs the whole forward/backward on its shard through
the HIP operators, and the ONLY exchange per step is DDP's all-reduce(mean) of the 3,494,549 fp32
gradients (13.98 MB — three ~5 MB buckets in gradient-completion order, ring all-reduces over xGMI that overlap the
remaining weight-gradient contractions: parallel.ddp_wrap) plus a 5-float all-reduce for logging.
"""
import torch
import torch.distributed as dist

from . import parallel


class Loss_manager():
    """Sums the loss terms the network returns (scripts/simple_train.py:23-54)."""
    keys = ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')

    def compute_loss(self, output_dict):
        total = None
        parts = {}
        like = next((v for v in (output_dict.get(k) for k in self.keys) if torch.is_tensor(v)), None)
        for k in self.keys:
            v = output_dict.get(k)
            if v is None:
                continue
            if not torch.is_tensor(v):          # e.g. smooth_loss == 0 (python int) when both smooth weights are <= 0
                v = like.new_tensor(float(v)) if like is not None else torch.as_tensor(float(v))
            if v.dim() > 0:                  # (a 0-dim term is its own mean: no reduction launch each way)
                v = v.mean()
            parts[k] = v.detach()
            total = v if total is None else total + v
        return total, parts


class Trainer():
    """net: a module with the UPFlow_net dict contract (input_dict -> output_dict with loss terms)."""

    def __init__(self, net, lr=1e-4, weight_decay=1e-4, scheduler_gamma=1.0, device=None, distributed=None, graph=False,
                 batch_check='collective', fused_adam=None):
        """fused_adam: the optimizer step as ONE multi-tensor kernel (torch.optim.Adam(fused=True): the same update formulas —
        Adam + amsgrad + L2 weight decay, scripts/simple_train.py:119-130 — in one pass over the 80 parameters instead of the
        15 `foreach` launches, 0.23 ms of a 11.8 ms step); default: on a GPU.
        batch_check (graph mode under DDP only): what happens when a rank is handed a batch that differs from the captured one
        (a last partial batch): 'collective' (default) — every step the ranks all-reduce one mismatch bit and, if ANY rank
        mismatches, ALL ranks take that step eagerly (same collective sequence everywhere: DDP's bucket all-reduce + the
        logging all-reduce); 'raise' — no per-step exchange, a mismatching rank raises ValueError.  A single process always
        decides locally (eager step + a warning)."""
        if batch_check not in ('collective', 'raise'):
            raise ValueError("batch_check must be 'collective' or 'raise', got %r" % (batch_check,))
        self.batch_check = batch_check
        self.device = device
        self.distributed = dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size() if self.distributed else 1
        self.rank = dist.get_rank() if self.distributed else 0
        self.raw_net = net if device is None else net.to(device)
        if self.distributed and graph and device is not None and torch.device(device).type == 'cuda':
            side = torch.cuda.Stream(device=device)              # (DDP + graph capture: construct on a side stream)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                self.net = parallel.ddp_wrap(self.raw_net, device)
            torch.cuda.current_stream(device).wait_stream(side)
        else:
            self.net = parallel.ddp_wrap(self.raw_net, device) if self.distributed else self.raw_net
        # graph=True: after `graph_warmup` eager steps the whole step (forward, losses, backward, Adam) is captured into ONE
        # hipGraph and replayed — a training step is ~2300 launches, i.e. 35-40 ms of python / ctypes / dispatcher time
        # that the GPU (22-30 ms of kernels once the convolutions run on the matrix cores) would otherwise wait for.
        # Needs fixed batch shapes; every libupflow_hip.so entry point only enqueues work, so the step is capturable.
        # Under DDP the capture follows PyTorch's whole-network recipe: DDP built on a side stream, 11 eager warm-up steps
        # (the reducer finalises its buckets), then fwd + bwd (incl. the bucket's RCCL all-reduce) + Adam captured.
        self.use_graph = bool(graph) and (device is not None) and torch.device(device).type == 'cuda'
        self.capture_fallback = False        # True once a capture failed and the trainer went back to eager steps
        self.capture_error = None            # 'ExceptionType: message' of that failure
        self._graph_keepalive = None         # pre-capture tensors whose addresses the captured graph reads (see _capture)
        self._mismatch_flag = None
        # graph mode: the learning rate is a DEVICE TENSOR — capturable Adam then reads it inside the captured step and the
        # scheduler updates it in place; a python float would be baked into the graph at capture time and every later
        # scheduler.step() silently ignored
        lr_arg = torch.tensor(float(lr), dtype=torch.float32, device=device) if self.use_graph else lr
        on_gpu = device is not None and torch.device(device).type == 'cuda'
        self.fused_adam = on_gpu if fused_adam is None else bool(fused_adam)
        self.optimizer = torch.optim.Adam([p for p in self.net.parameters() if p.requires_grad], lr=lr_arg, amsgrad=True,
                                          weight_decay=weight_decay, capturable=self.use_graph, fused=self.fused_adam)
        # (the `optimizer` setter below registered the version-counter hook on it and took its parameter list)
        self.graph_warmup = 11 if self.distributed else 3
        self._graph = None
        self._static = None
        self._static_stats = None
        self._eager_steps = 0
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=scheduler_gamma)
        self.loss_manager = Loss_manager()

    @property
    def optimizer(self):
        return self._optimizer

    @optimizer.setter
    def optimizer(self, opt):
        """The packed 16-bit weight copies are keyed on the parameters' autograd version counters, which torch's FUSED optimizers
        do not advance: a per-optimizer post-step hook does (ops.register_version_hook; custom training loops around a fused
        optimizer need the same hook).  ANY optimizer assigned to the trainer — the one built in __init__ or a replacement
        (`tr.optimizer = torch.optim.Adam(..., fused=True)`) — gets the hook, and the replay path's parameter list follows it
        (a replaced fused optimizer had neither, and the step kept multiplying by the step-0 packed weights)."""
        from . import ops as _ops
        self._optimizer = opt
        _ops.register_version_hook(opt)
        self._params_flat = [p for g in opt.param_groups for p in g['params']]

    def shard(self, batch):
        """This rank's contiguous-strided slice of a GLOBAL batch dict (DistributedSampler-style)."""
        if self.world == 1:
            return batch
        n = next(v for v in batch.values() if torch.is_tensor(v)).shape[0]
        idx = parallel.shard_indices(n, self.rank, self.world)
        return {k: (v[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}

    def _step_body(self, batch):
        batch = dict(batch)
        batch['if_loss'] = True
        out = self.net(batch)
        loss, parts = self.loss_manager.compute_loss(out)
        loss.backward()                      # DDP overlaps the gradient all-reduce with the rest of backward
        self.optimizer.step()
        # (torch._fused_adam_ updates the parameters WITHOUT advancing their autograd version counters, which is what the packed
        # 16-bit weight copies of the convolution path are keyed on — ops.conv_pack_from_master, pwc_modules._PackedConv3x3: the
        # next forward, and at capture time the captured step, would keep multiplying by the weights of the step before; found by
        # test_config3_full_size_step_graphed_equals_eager_and_bf16_tracks_fp32.  The post-step hook registered in __init__ on
        # THIS optimizer — ops.register_version_hook — has advanced them by now.)
        self._names = ['loss'] + sorted(parts)
        stats = torch.stack([loss.detach().float()] + [parts[k].float() for k in sorted(parts)])
        if self.distributed:                 # loss terms averaged over ranks (one 5-float all-reduce, for logging)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            stats = stats / self.world
        return stats

    def _capture(self, batch):
        dev = torch.device(self.device)
        self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        self.optimizer.zero_grad(set_to_none=True)
        from . import ops
        from .utils import loss as loss_mod
        mark = ops.train_caches_mark()
        g = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread polls its events while this thread captures; under the
        # default (global) capture mode that hipEventQuery is an error that aborts the process
        try:
            # (also when this trainer is not distributed but the process holds a process group: its threads are there all the same)
            with parallel.no_gc_during_capture(), \
                    torch.cuda.graph(g, capture_error_mode='thread_local' if (self.distributed or (dist.is_available() and dist.is_initialized())) else 'global'):
                self._static_stats = self._step_body(self._static)
        finally:
            # packed-weight cache entries made DURING the capture point into graph-pool memory whose packing kernels were
            # only RECORDED: an eager step that found them would multiply by garbage -> dropped.  Entries that
            # existed BEFORE it and were cache hits inside it (the zero-bias operand of the data-gradient convolutions, packs
            # of frozen parameters) and the loss module's constants have their eager-pool addresses baked into the graph:
            # they stay cached AND are pinned here for the graph's lifetime (dropping them was a use-after-free).
            # (+ the packed operands of the inference-style holders, pwc_modules._PackedConv*: forwards under no_grad inside the step
            #  read them, and a holder re-packs — frees — its tensor when its parameter's version moves; round 4)
            from .model.pwc_modules import packed_operands
            self._graph_keepalive = ops.train_caches_after_capture(mark) + loss_mod.cached_constants() + packed_operands(self.raw_net)
        self._graph = g
        torch.cuda.synchronize(dev)

    def step(self, batch, sync_stats=True):
        """One optimisation step on this rank's shard; returns the loss terms averaged over ranks
        (sync_stats=False: the device tensor of the terms, no host synchronisation)."""
        self.net.train()
        if self.use_graph and self._graph is not None and not self._replay_agreed(batch):
            import warnings
            warnings.warn('batch differs from the captured one (keys / shapes / non-tensor values)%s: this step runs eagerly'
                          % (' on at least one rank' if self.distributed else ''))
            self.optimizer.zero_grad(set_to_none=True)
            stats = self._step_body(batch)
        elif self.use_graph and self._graph is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self._static[k].copy_(v, non_blocking=True)
            self._graph.replay()
            stats = self._static_stats
            # the replayed Adam kernel has changed every parameter ON THE DEVICE; no host code ran, so nothing advanced their
            # autograd version counters — which is what the packed weight copies of the inference-style holders
            # (pwc_modules._PackedConv*: a validation forward between training steps) and GraphedInference.check_weights are keyed
            # on (eval, replay, eval would have multiplied by the weights of the first eval).  Host-only, microseconds.
            torch.autograd.graph.increment_version(self._params_flat)
        else:
            self.optimizer.zero_grad(set_to_none=True)
            stats = self._step_body(batch)
            self._eager_steps += 1
            if self.use_graph and self._eager_steps >= self.graph_warmup:
                err = None
                try:
                    self._capture(batch)         # (the capture itself does not execute: this step already ran eagerly)
                except Exception as e:           # e.g. a collective library that cannot be captured: stay eager, say so
                    err = '%s: %s' % (type(e).__name__, e)
                # the decision is COLLECTIVE: a rank whose capture failed steps eagerly from here on, and would pair
                # its python-issued collectives with the other ranks' captured ones (and skip the per-step mismatch-bit exchange
                # they issue) -> every rank falls back if any rank failed
                failed_somewhere = self._any_rank(err is not None)
                if failed_somewhere:
                    import warnings
                    err = err or 'the capture failed on another rank'
                    warnings.warn('hipGraph capture of the training step failed (%s); continuing with eager steps' % err)
                    self.use_graph = False
                    self.capture_fallback = True
                    self.capture_error = err
                    self._graph = None
                    self._static = self._static_stats = None
                    self._graph_keepalive = None
                    torch.cuda.synchronize(torch.device(self.device))
        if not sync_stats:
            return stats
        return {k: float(v) for k, v in zip(self._names, stats.cpu())}

    def _any_rank(self, flag):
        """True if `flag` is true on ANY rank (one MAX all-reduce of a bit under DDP; the local value in a single process)."""
        if not self.distributed or self.world == 1:
            return bool(flag)
        if self._mismatch_flag is None:
            self._mismatch_flag = torch.zeros(1, dtype=torch.int32, device=self.device if self.device is not None else 'cpu')
        self._mismatch_flag.fill_(1 if flag else 0)
        dist.all_reduce(self._mismatch_flag, op=dist.ReduceOp.MAX)
        return int(self._mismatch_flag.item()) != 0

    def _replay_agreed(self, batch):
        """True: replay the captured step; False: this step runs eagerly.  A single process decides by its own batch.  Under DDP
        the decision must be the SAME on every rank — a rank that stepped eagerly (DDP's bucket all-reduce issued from python)
        while the others replay their graphs (the same all-reduce inside the graph, then the logging all-reduce) would pair
        mismatched collectives or hang: the ranks exchange one bit (MAX) per step, or, with
        batch_check='raise', a mismatching rank raises instead."""
        ok = self._matches_static(batch)
        if not self.distributed or self.world == 1:
            return ok
        if self.batch_check == 'raise':
            if not ok:
                raise ValueError('rank %d: batch differs from the captured one (keys / shapes / non-tensor values) — under DDP with '
                                 "batch_check='raise' every rank must feed the captured shapes (drop or pad the last partial batch)" % self.rank)
            return True
        return not self._any_rank(not ok)

    def _matches_static(self, batch):
        """The captured step replays on the tensors it was captured with: same keys, same shapes / dtypes, same non-tensor
        values (a last partial batch of an epoch, a changed flag ... must not be copied into them)."""
        st = self._static
        if set(batch) - {'if_loss'} != set(st) - {'if_loss'}:
            return False
        for k, v in batch.items():
            if k == 'if_loss':
                continue
            if torch.is_tensor(v):
                if not torch.is_tensor(st[k]) or v.shape != st[k].shape or v.dtype != st[k].dtype:
                    return False
            elif torch.is_tensor(st[k]) or v != st[k]:
                return False
        return True

    def end_epoch(self):
        self.scheduler.step()


def synthetic_train_batch(B, crop_hw=(256, 832), raw_hw=(288, 864), seed=0, device='cpu', dtype=torch.float32):
    """A KITTI-shaped synthetic batch (dataset/kitti_dataset.py:268-342 yields crops of 256x832 plus the
    un-cropped frames and the crop offset `start`)."""
    g = torch.Generator().manual_seed(4000 + seed)
    H, W = raw_hw
    h, w = crop_hw
    base = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=g)
    big = torch.nn.functional.interpolate(base, size=(H + 8, W + 8), mode='bicubic', align_corners=True) - 0.45
    im1 = big[:, :, 4:4 + H, 4:4 + W].contiguous()
    im2 = big[:, :, 4:4 + H, 2:2 + W].contiguous()                      # 2-px horizontal motion
    sy, sx = (H - h) // 2, (W - w) // 2
    start = torch.tensor([sx, sy], dtype=torch.float32).view(1, 2, 1, 1).repeat(B, 1, 1, 1)
    batch = {'im1': im1[:, :, sy:sy + h, sx:sx + w].contiguous(), 'im2': im2[:, :, sy:sy + h, sx:sx + w].contiguous(),
             'im1_raw': im1, 'im2_raw': im2, 'start': start}
    return {k: v.to(device=device, dtype=dtype if k != 'start' else torch.float32) for k, v in batch.items()}
