#!/usr/bin/env python3
"""bench.py — frame-pairs/sec of the UPFlow inference hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N>1 without a launcher: bench.py re-executes itself under torch.distributed.run (one rank per GPU, RCCL);
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          (the driver's form: RANK/LOCAL_RANK/WORLD_SIZE from the env)

A "step" is one UPFlow_net inference forward (both flow directions + occlusion masks, SGU on, flags
of the reference's test.py:22-30) over one batch of synthetic frame pairs, inputs already resident in
HBM.  Workload = BASELINE config 2: 384x1280, bf16, batch 4 per GPU.  Image pairs shard across ranks
with no data-path collective (replicas, weak scaling); value = pairs all ranks processed / max-over-
ranks time of exactly K steps bracketed by barrier + synchronize.

The JSON line also carries
  roofline     — the dominant hand-written kernel (81-neighbour cost volume at the 1/4-res level of
                 this workload): algorithmic bytes s*B*H*W*(2C+81) / average kernel duration measured
                 with HIP events recorded around each launch on its stream, vs the 8 TB/s HBM peak;
  cpu_baseline — the reference's pure-PyTorch fallback path (utils/pytorch_correlation.py algorithm,
                 restated in oracle/) timed on this box's host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (B per GPU, H, W, dtype)
    'config2': (4, 384, 1280, 'bf16'),
    'config4': (8, 448, 1024, 'fp16'),
    'config5': (1, 960, 2880, 'bf16'),
    'kitti_native': (4, 375, 1242, 'bf16'),      # not a BASELINE config: un-padded KITTI frames, every pyramid level ragged
}
DT = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_froze_pwc': False, 'if_sgu_upsample': True}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PEAK_SUSTAINED_TFLOPS = 1420.0  # measured: the vendor bf16 GEMM's best case on random data on this part (profiles/r06_gemm_ceiling.txt)


def build_net(dtype, device, hip_pyramid_convs=True, fp32_conv='hip_x3', pyramid_dtype=None):
    from upflow_pytorch_amd import synthetic as _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    conf.update(dict(FLAGS, hip_pyramid_convs=hip_pyramid_convs, fp32_conv=fp32_conv), verbose=False)
    torch.manual_seed(0)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))      # random-init weights of the architecture
    if pyramid_dtype is not None:
        return net.to(device).to_inference(dtype, DT[pyramid_dtype])
    return net.to(device).to(dtype).eval()


def roofline_probe(B, H, W, dtype, device, feature_dtype=None):
    """Dominant hand-written kernel: corr81 forward at the 1/4-resolution level (C=32) of this workload, with the shape
    of the launch the model makes there ([2B,32,H/4,W/4]: both flow directions of the B frame pairs in one launch).

    `achieved` / `frac` = algorithmic bytes s*B*H*W*(2C+81) / the average duration of 200 BACK-TO-BACK launches, each
    bracketed by its own pair of HIP events recorded on the launch stream (hipExtLaunchKernel start/stop events) — the
    inputs of a back-to-back launch are resident in the 256 MB infinity cache, as they are in the pipeline where the
    normalisation kernel has just written them.  `frac_cold` is the same launch after a 512 MB fill has evicted L2 and the
    infinity cache (every byte from HBM).  rocprofv3 --kernel-trace of this very command reports the same launches
    (profiles/r02_*: its per-kernel average sits ~1 us above the event figure, both are committed)."""
    from upflow_pytorch_amd import ops
    C, h, w = 32, (H + 3) // 4, (W + 3) // 4
    B = 2 * B            # the launch the model makes: both flow directions stacked along the batch (UPFlow_net._forward_stacked)
    g = torch.Generator(device='cpu').manual_seed(2004)
    # (the [features; warped] pair buffer of the level as the step allocates it: row-pitched when the level width is ragged)
    # feature_dtype: the pyramid's type when it differs from the decoder's (`pyramid_dtype`: fp16 features, bf16 cost volume)
    fdt = feature_dtype if (feature_dtype is not None and dtype != torch.float32) else dtype
    pair = ops.empty_nchw((2, B, C, h, w), fdt, device)
    pair[0].copy_(torch.randn(B, C, h, w, generator=g).to(device))
    pair[1].copy_(torch.randn(B, C, h, w, generator=g).to(device))
    f1, f2 = pair[0], pair[1]
    out = torch.empty(B, 81, h, w, device=device, dtype=dtype)
    # the kernel INSIDE the timed step is the variant whose loader normalises the features (upf_corr81_norm_forward: 16-bit
    # inference); the plain variant (training, fp32) is reported beside it
    norm = dtype != torch.float32 and ops.corr81_norm_supported(f1)
    # ... and at the levels whose flow estimator runs in the channel-octet layout (this one, at config 2) it stores octets
    from upflow_pytorch_amd.model.pwc_modules import c8_level_ok
    c8 = norm and c8_level_ok(B, h, w, dtype)
    pitched = ops.nchw_pitch(f1) != w
    if pitched and not c8:
        f1, f2 = f1.contiguous(), f2.contiguous()                            # (the timed NCHW-output helpers take contiguous features)
    if fdt != dtype and not c8:
        fdt = dtype                                   # (only the octet form has a timed mixed entry: time the single-type launch)
        pair = pair.to(dtype)
        f1, f2 = pair[0], pair[1]
    if c8:
        out8 = ops.c8_empty(B, 88, h, w, dtype, device)
        timed = lambda f1_, f2_, out_, slope, nrep: ops.corr81_norm_forward_c8_timed(f1_, f2_, out8, slope, nrep=nrep)
    else:
        timed = ops.corr81_norm_forward_timed if norm else ops.corr81_forward_timed
    timed(f1, f2, out, 0.1, nrep=20)                                         # warm
    avg_us, min_us = timed(f1, f2, out, 0.1, nrep=200)
    fc1, fc2 = (f1.contiguous(), f2.contiguous()) if pitched else (f1, f2)
    if fdt != dtype:
        fc1, fc2 = fc1.to(dtype), fc2.to(dtype)       # (the side variants are single-type launches)
    nchw_norm_us = ops.corr81_norm_forward_timed(fc1, fc2, out, 0.1, nrep=200)[0] if c8 else None
    plain_us = ops.corr81_forward_timed(fc1, fc2, out, 0.1, nrep=200)[0] if norm else avg_us
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=device)
    cold = []
    for _ in range(30):
        flush.fill_(1)
        cold.append(timed(f1, f2, out, 0.1, nrep=1)[0])
    del flush
    cold_us = sum(cold) / len(cold)
    s = f1.element_size()
    alg_bytes = s * B * h * w * (2 * C + 81)
    achieved = alg_bytes / (avg_us * 1e-6) / 1e9
    # HBM traffic per launch from the PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) cannot be
    # collected from inside this process; it is the committed rocprofv3 measurement of the same launch
    # (profiles/*_pmc.json, separate --pmc passes) when shape and dtype match.
    traffic = None
    dn = {torch.bfloat16: 'bf16', torch.float16: 'fp16'}.get(dtype, 'fp32')
    for name in sorted(os.listdir(os.path.join(ROOT, 'profiles')), reverse=True):
        if not (name.endswith('_pmc.json') and 'corr81' in name):
            continue
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))['summary']
            if pmc['shape'] == [B, C, h, w] and pmc['dtype'] == dn and pmc.get('variant', 'plain') == ('norm_c8' if c8 else 'norm' if norm else 'plain'):
                traffic = int(pmc['traffic_bytes'])
                break
        except Exception:
            pass
    kname = 'corr81_fwd_kernel' if dtype == torch.float32 else 'corr81_allc_kernel<8x32 tile>' if not norm else \
        ('corr81_allc_kernel<8x32 tile, NORM: normalisation fused into the loader, OC8: octet output> (the launch inside the step)' if c8 else
         'corr81_allc_kernel<8x32 tile, NORM: normalisation fused into the loader> (the launch inside the step)')
    # (octet output: the kernel STORES 88 positions per pixel — the 7 zero positions of the 11th octet — while `achieved` prices
    # the 81 algorithmic channels; stored / algorithmic bytes = (2C + 88) / (2C + 81))
    extra = {'nchw_output_variant_us': round(nchw_norm_us, 2), 'stored_over_algorithmic_bytes': round((2 * C + 88) / (2 * C + 81), 4)} if c8 else {}
    if pitched:
        extra['feature_row_pitch'] = ops.nchw_pitch(f1)
    if fdt != dtype:
        kname += ' [features %s -> cost volume %s]' % (str(fdt).replace('torch.', ''), str(dtype).replace('torch.', ''))
    return {'bound': 'hbm', 'kernel': kname, 'shape': [B, C, h, w], **extra,
            'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic, 'algorithmic_bytes': alg_bytes, 'avg_kernel_us': round(avg_us, 2), 'min_kernel_us': round(min_us, 2),
            'timing': 'HIP events around each of 200 back-to-back launches (inputs resident in the infinity cache)',
            'cold_kernel_us': round(cold_us, 2), 'frac_cold': round(alg_bytes / (cold_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            'plain_variant_us': round(plain_us, 2), 'plain_variant_frac': round(alg_bytes / (plain_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}


def conv_roofline_probe(B, H, W, dtype, device, fp32_conv='hip_x3'):
    """The kernel a step spends most of its time in: the matrix-core convolution, measured on its largest launch —
    the context network's first layer (565 -> 128 channels, 3x3) at the 1/4-resolution level, both flow directions
    stacked (2B items).  MFMA-bound: algorithmic flop 2*9*Cin*Cout*N*H*W over the average of `nrep` back-to-back
    launches between two HIP events on the launch stream, against the dense bf16/fp16 MFMA peak (2.5 PFLOP/s)."""
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.model import pwc_modules
    x3 = dtype == torch.float32
    if x3 and fp32_conv == 'miopen':
        return None                                                          # (PyTorch-ROCm's convolutions: not this build's kernel)
    if x3:
        with pwc_modules.fp32_conv_mode(fp32_conv):                          # (sets the kernel's product mode: hip_x3 / hip_x3s)
            return _conv_roofline_probe(B, H, W, dtype, device, True)
    return _conv_roofline_probe(B, H, W, dtype, device, False)


def _conv_roofline_probe(B, H, W, dtype, device, x3):
    from upflow_pytorch_amd import ops
    N, Cin, Cout, h, w = 2 * B, 565, 128, (H + 3) // 4, (W + 3) // 4
    g = torch.Generator(device='cpu').manual_seed(2005)
    x = torch.randn(N, Cin, h, w, generator=g).to(device).to(dtype)
    wgt = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(device).to(dtype)
    bias = torch.zeros(Cout, device=device)
    y = torch.empty(N, Cout, h, w, device=device, dtype=dtype)
    packed = ops.conv3x3_pack(wgt)
    for _ in range(5):
        ops.conv3x3_forward_raw(x, packed, bias, y, 1, 0.1)
    nrep = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep):
        ops.conv3x3_forward_raw(x, packed, bias, y, 1, 0.1)
    e1.record()
    torch.cuda.synchronize(device)
    avg_us = e0.elapsed_time(e1) * 1e3 / nrep
    flop = 2.0 * 9 * Cin * Cout * N * h * w
    achieved = flop / (avg_us * 1e-6) / 1e12
    if x3:
        # the split-precision kernel issues THREE fp16 matrix products per operand pair: `achieved` / `frac` price the ALGORITHMIC flop
        # (what the layer computes) against the dense fp16 peak; `issued_frac` = 3x that, the share of the matrix pipe it occupies
        return {'bound': 'mfma', 'kernel': 'conv_x3_kernel<MTW=2> (split precision, fp16 hi/lo x 3; context network layer 1, 565->128, 3x3)',
                'shape': [N, Cin, h, w], 'achieved': round(achieved, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(achieved / 2500.0, 4),
                'issued_frac': round(3 * achieved / 2500.0, 4), 'traffic': None, 'algorithmic_flop': flop, 'avg_kernel_us': round(avg_us, 2)}
    nchw_us, nchw_tf = avg_us, achieved
    kname = 'conv_kernel<MTW=4> (context network layer 1, 565->128, 3x3)'
    from upflow_pytorch_amd.model.pwc_modules import c8_level_ok
    if c8_level_ok(N, h, w, dtype):
        # the launch the STEP makes at this level: operands in the channel-octet layout (LDS-DMA staging, octet epilogue), the 565 input
        # channels in 72 octets as the estimator's buffer holds them (csrc/conv_c8.hip); the NCHW form above is reported beside it
        n8 = (Cin + 7) // 8
        x8 = ops.to_c8(x)
        y8 = ops.c8_empty(N, Cout, h, w, dtype, device)
        packed8 = ops.conv_c8_pack(wgt, list(range(Cin)) + [-1] * (n8 * 8 - Cin))
        for _ in range(5):
            ops.conv_c8_forward_raw(x8, None, packed8, bias, y8, 1, 0.1)
        e0.record()
        for _ in range(nrep):
            ops.conv_c8_forward_raw(x8, None, packed8, bias, y8, 1, 0.1)
        e1.record()
        torch.cuda.synchronize(device)
        avg_us = e0.elapsed_time(e1) * 1e3 / nrep
        achieved = flop / (avg_us * 1e-6) / 1e12
        kname = 'conv_kernel<MTW=4, octet operands> (context network layer 1, 565->128, 3x3: the launch inside the step)'
    return {'bound': 'mfma', 'kernel': kname, 'shape': [N, Cin, h, w],
            'achieved': round(achieved, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(achieved / 2500.0, 4), 'traffic': None,
            'algorithmic_flop': flop, 'avg_kernel_us': round(avg_us, 2), 'nchw_form_us': round(nchw_us, 2), 'nchw_form_tflops': round(nchw_tf, 1),
            # what the vendor's bf16 GEMM sustains on this part on random data in its best case (hipBLASLt 8192^3: 1410-1433 TFLOP/s at
            # 1320 W / 1.97 GHz; at this layer's own M x N x K it holds 510-660): tools/gemm_ceiling.py, DESIGN.md 4.2
            'peak_sustained': PEAK_SUSTAINED_TFLOPS, 'frac_of_sustained': round(achieved / PEAK_SUSTAINED_TFLOPS, 4),
            'peak_sustained_source': 'profiles/r06_gemm_ceiling.txt'}


def conv_flop_per_step(net, B, H, W):
    """Algorithmic flop (2 * k*k * Cin * Cout * output pixels) of every convolution one inference step runs, from the module
    shapes and the schedule of UPFlow_net._forward_stacked (both directions stacked: 2B items): six pyramid stages, the 1x1
    projections / flow estimator / context network at the five decoder levels, the SGU estimator at levels 1-4 and at the final
    up-sampling, the SGU guidance stem on the frames."""
    nb = 2 * B

    def seq_flop(seq, h, w):
        c = seq[0]
        s = c.stride[0]
        ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
        return 2.0 * c.kernel_size[0] * c.kernel_size[1] * c.in_channels * c.out_channels * ho * wo * nb, ho, wo
    total, parts = 0.0, {}

    def add(name, f):
        nonlocal total
        total += f
        parts[name] = parts.get(name, 0.0) + f
    h, w = H, W
    sizes = []
    for stage in net.feature_pyramid_extractor.convs:
        for seq in stage:
            f, h, w = seq_flop(seq, h, w)
            add('pyramid', f)
        sizes.append((h, w))
    levels = sizes[::-1][:net.output_level + 1]                      # coarsest first
    for lvl, (h, w) in enumerate(levels):
        add('conv_1x1', seq_flop(net.conv_1x1[lvl], h, w)[0])
        for name in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'conv_last'):
            add('estimator', seq_flop(getattr(net.flow_estimators, name), h, w)[0])
        for seq in net.context_networks.convs:
            add('context', seq_flop(seq, h, w)[0])
    if net.sgi_model is not None:
        em = net.sgi_model.dense_estimator_mask
        for (h, w) in levels[1:] + [levels[-1]]:
            for name in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'conv_last'):
                add('sgu_estimator', seq_flop(getattr(em, name), h, w)[0])
        h, w = H, W
        for seq in net.sgi_model.upsample_output_conv:
            f, h, w = seq_flop(seq, h, w)
            add('sgu_stem', f)
    return total, parts


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline():
    """The reference's CPU fallback (pure-PyTorch unfold correlation inside the full fp32 forward), as restated in
    oracle/ (kind "port"), on this box's host cores — BASELINE.md §3: full UPFlow_net forward at 384x1280 B=1 (the
    metric's shape; `value`) and 256x256 B=1 (config 1), median of 4 timed runs after 1 warm-up, the share of the time
    spent inside the 10 fallback-correlation calls, and the correlation alone at the five pyramid-level shapes."""
    from upflow_pytorch_amd import synthetic as _weights
    from oracle import net as onet
    from oracle import ops as oops
    sd = _weights.make_state_dict(0, head_scale=0.1)
    spent = [0.0]
    inner = onet._corr

    def timed_corr(a, b, corr):
        t = time.perf_counter()
        r = inner(a, b, corr)
        spent[0] += time.perf_counter() - t
        return r
    res = {}
    onet._corr = timed_corr
    try:
        with torch.no_grad():
            for name, (H, W, cid) in (('256x256', (256, 256, 1)), ('384x1280', (384, 1280, 2))):
                im1, im2 = _weights.make_images(cid, 1, H, W)
                times, shares = [], []
                for it in range(5):                          # (a bounded sample: ~25 s of host time in all)
                    spent[0] = 0.0
                    t0 = time.perf_counter()
                    onet.forward(sd, im1, im2, mask_mode='literal', corr='unfold')
                    dt = time.perf_counter() - t0
                    if it >= 1:
                        times.append(dt)
                        shares.append(spent[0] / dt)
                res[name] = {'s_per_pair': round(_median(times), 4), 'frame_pairs_per_s': round(1.0 / _median(times), 5),
                             'correlation_share': round(_median(shares), 3)}
            levels = {}
            for C, h, w in ((196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)):
                g = torch.Generator().manual_seed(2000 + C)
                a, b = torch.randn(1, C, h, w, generator=g), torch.randn(1, C, h, w, generator=g)
                ts = []
                for it in range(7):
                    t0 = time.perf_counter()
                    oops.corr81_unfold(a, b)
                    if it >= 2:
                        ts.append(time.perf_counter() - t0)
                levels['%dx%dx%d' % (C, h, w)] = round(_median(ts) * 1e3, 3)
    finally:
        onet._corr = inner
    main_ = res['384x1280']
    return {'value': main_['frame_pairs_per_s'], 'unit': 'frame-pairs/s', 'cores': torch.get_num_threads(),
            'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': 'full UPFlow_net fp32 forward of ONE 384x1280 frame pair with the unfold-based fallback correlation '
                      '(utils/pytorch_correlation.py:27-50 restated in oracle/): median of 4 timed runs after 1 warm-up, '
                      '%.2f s per pair, %.0f %% of it inside the 10 correlation calls' % (main_['s_per_pair'], 100 * main_['correlation_share']),
            'shapes': res, 'correlation_alone_ms': levels}


TRAIN_FLAGS = {'photo_loss_census_weight': 1, 'multi_scale_distillation_weight': 1, 'multi_scale_distillation_style': 'upup',
               'multi_scale_distillation_occ': True, 'smooth_order_1_weight': 1, 'if_use_boundary_warp': True}


def train_main(args, rank, world, device):
    """BASELINE config 3: unsupervised training step, global batch 4*N, DDP gradient all-reduce over RCCL."""
    from upflow_pytorch_amd import synthetic as _weights
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
    if os.environ.get('UPF_MIOPEN_BENCHMARK'):
        torch.backends.cudnn.benchmark = True          # MIOpen exhaustive solver search for the fp32 convolutions
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(TRAIN_FLAGS)
    dname = args.dtype or 'bf16'            # bf16 / fp16 (the default): activations in that type, fp32 master weights, forward /
    d['train_conv_dtype'] = dname           # data gradient / weight gradient on the MFMA kernels;  --dtype fp32: every convolution
    conf.update(d, verbose=False)           # PyTorch-ROCm (the parity mode, ~4x slower)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    tr = Trainer(net, device=device, graph=not args.no_graph)
    B = 4
    batch = synthetic_train_batch(B, seed=rank, device=device)                    # a different shard per rank

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)
    for _ in range(max(args.warmup, tr.graph_warmup + 1 if tr.use_graph else 0)):
        tr.step(batch)
    for _ in range(int(args.ramp_seconds * 60)):            # clock ramp (see the inference loop); a FIXED count: every step
        tr.step(batch)                                      # is a collective under DDP, so all ranks must run the same number
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = tr.step(batch, sync_stats=False)     # (no host round trip inside the timed region; the barrier below synchronises)
    torch.cuda.synchronize(device)
    own = time.perf_counter() - t0                   # (under DDP every step ends in a collective: the ranks' own times differ by their last step only)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device)
    stats = {k: float(v) for k, v in zip(tr._names, stats.cpu())}
    spread = rank_spread(own, args.steps, device)
    # the ONE exchange step of the path, timed on its own (outside the timed region): an all-reduce of a gradient-sized
    # fp32 buffer on the process group DDP uses, HIP events around 10 back-to-back calls
    allreduce_ms = None
    if world > 1:
        n_grad = sum(p.numel() for p in tr.raw_net.parameters() if p.requires_grad)
        buf = torch.zeros(n_grad, dtype=torch.float32, device=device)
        for _ in range(3):
            torch.distributed.all_reduce(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record()
        for _ in range(10):
            torch.distributed.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize(device)
        allreduce_ms = parallel.max_over_ranks(e0.elapsed_time(e1) / 10, device)
    if rank == 0:
        print(json.dumps({
            'metric': 'training frame-pairs/sec (unsupervised step, 256x832 crops)', 'value': round(world * B * args.steps / elapsed, 3),
            'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dname, 'data': 'synthetic',
            'config': {'workload': 'config3: photometric + smooth + census + pyramid-distillation loss, fwd+bwd+Adam(amsgrad), '
                                   '256x832 crops of 288x864 frames, batch 4 per GPU', 'global_batch': world * B,
                       'parallelism': 'dp%d (DDP, %d gradient bucket(s) in completion order, RCCL all-reduce inside the captured step)' % (world, max(1, len(parallel.ddp_bucket_bytes(tr.net)))),
                       'gradient_buckets_bytes': parallel.ddp_bucket_bytes(tr.net) or None, 'ranks': world, 'rank_ms_per_step': spread,
                       'hip_graph': tr.use_graph,
                       'capture_fallback': tr.capture_fallback,      # True: the hipGraph capture failed and the steps ran eagerly
                       'optimizer': 'torch.optim.Adam(amsgrad, weight_decay 1e-4, %s)' % ('fused: one multi-tensor kernel' if tr.fused_adam else 'foreach'),
                       'backend': (torch.distributed.get_backend() + ' (RCCL)') if world > 1 else None,
                       'gradient_allreduce_ms': None if allreduce_ms is None else round(allreduce_ms, 4),
                       'gradient_bytes': 4 * sum(p.numel() for p in tr.raw_net.parameters() if p.requires_grad)},
            'roofline_train': train_roofline(tr.raw_net, B, 256, 832, elapsed / args.steps * 1e3) if dname != 'fp32' else None,
            'final_loss': stats}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def train_probe(device, steps=30):
    """BASELINE config 3 at N = 1 as an extra object of the default line (so that the driver's own run carries a training
    number): one unsupervised training step (forward, photometric / smooth / census / pyramid-distillation losses, backward,
    Adam(amsgrad)) on the matrix cores (bf16 activations, fp32 master weights) inside one hipGraph — `--mode train` is the
    full bench of the same step (any N, DDP)."""
    from upflow_pytorch_amd import synthetic as _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
    try:
        conf = UPFlow_net.config()
        d = dict(FLAGS)
        d.update(TRAIN_FLAGS)
        d['train_conv_dtype'] = 'bf16'
        conf.update(d, verbose=False)
        net = conf()
        net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
        tr = Trainer(net, device=device, graph=True)
        batch = synthetic_train_batch(4, seed=0, device=device)
        for _ in range(tr.graph_warmup + 1 + 20):
            tr.step(batch)
        torch.cuda.synchronize(device)
        wins = []
        for _w in range(3):                                  # three windows of `steps` steps: the median
            t0 = time.perf_counter()
            for _ in range(steps):
                stats = tr.step(batch, sync_stats=False)
            torch.cuda.synchronize(device)
            wins.append((time.perf_counter() - t0) / steps * 1e3)
        wins.sort()
        ms = wins[1]
        loss = float(stats.cpu()[tr._names.index('loss')]) if 'loss' in tr._names else None
        return {'workload': 'config3: unsupervised training step, 256x832 crops, batch 4, bf16 activations / fp32 master weights, '
                            'forward + losses + backward + Adam(amsgrad) in one hipGraph', 'dtype': 'bf16', 'ms_per_step': round(ms, 3),
                'ms_per_step_min': round(wins[0], 3), 'ms_per_step_max': round(wins[2], 3), 'windows': 3,
                'frame_pairs_per_s': round(4e3 / ms, 2), 'steps': steps, 'hip_graph': tr.use_graph, 'capture_fallback': tr.capture_fallback, 'final_loss': loss,
                'roofline_train': train_roofline(tr.raw_net, 4, 256, 832, ms)}
    except Exception as e:                                   # (an extra: it must never take the headline line down)
        return {'error': '%s: %s' % (type(e).__name__, e)}


def eval_probe(net, H, W, device, iters=40):
    """The reference's evaluation workload (test.py:40-47: batch 1, one frame pair at a time, inputs resident in HBM): ms per
    pair of upflow_pytorch_amd.test.Test_model.eval_forward through runtime.ShapeCachedInference (one captured graph per frame
    size, ONE step in flight: the latency a caller of eval_forward sees) against the eager forward (one ctypes launch per kernel)."""
    from upflow_pytorch_amd import synthetic as _weights
    from upflow_pytorch_amd.test import Test_model
    try:
        a, b = _weights.make_images(77, 1, H, W)
        a, b = a.to(device), b.to(device)
        res = {}
        for name, graph in (('graph', True), ('eager', False)):
            tm = Test_model(pretrain_path=None, dtype=None, graph=graph, device=device, net=net)      # (dtype None: the supplied network keeps its — possibly mixed — types)
            for _ in range(5):
                tm.eval_forward(a, b, 0)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(iters):
                tm.eval_forward(a, b, 0)
            torch.cuda.synchronize(device)
            res[name] = (time.perf_counter() - t0) / iters * 1e3
        # ... and with four pairs in flight (runtime.PipelinedEvaluation behind Test_model(streams=4).eval_forward_stream): what
        # Evaluation_bench uses when the test model offers it
        tm = Test_model(pretrain_path=None, dtype=None, device=device, net=net, streams=4)
        for _ in tm.eval_forward_stream([(a, b)] * 8):
            pass
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in tm.eval_forward_stream([(a, b)] * (4 * iters)):
            pass
        torch.cuda.synchronize(device)
        res['pipelined'] = (time.perf_counter() - t0) / (4 * iters) * 1e3
        return {'workload': 'Test_model.eval_forward, batch 1, %dx%d, one pair at a time (test.py:40-47)' % (H, W),
                'graph_ms_per_pair': round(res['graph'], 3), 'eager_ms_per_pair': round(res['eager'], 3),
                'graph_pairs_per_s': round(1e3 / res['graph'], 1), 'iters': iters,
                'four_pairs_in_flight_ms_per_pair': round(res['pipelined'], 3), 'four_pairs_in_flight_pairs_per_s': round(1e3 / res['pipelined'], 1)}
    except Exception as e:                                   # (an extra: it must never take the headline line down)
        return {'error': '%s: %s' % (type(e).__name__, e)}


def epe_probe(net, streams, graph, device):
    """The second half of BASELINE's metric — "EPE vs reference" (mean over pixels of |flow - flow_ref|_2: the definition of
    /root/reference/dataset/kitti_dataset.py:464-475 with mask = 1) — produced IN THE RUN by the path the headline times: the SAME
    network object and dtypes, the same 384x1280 batch-4 captured step, the same number of steps in flight, every slot replayed
    concurrently.  What it is compared with is DATA that travels with the repository: tests/golden/net_384x1280_hs1_robust.npz, the
    REFERENCE's own fp32 output (generated by tests/golden/make_golden.py, which imports /root/reference) on two synthetic frame
    pairs with full-scale prediction heads (mean |flow| 15.6 px, p99 34, max 56 — KITTI-sized motion; the headline's timing
    weights keep the heads at 0.1 scale) under the exact-predicate ('robust') warp mask, the protocol in which the reference is
    not chaotic against itself.  So for the measurement the network object gets the fixture's weights and
    mask mode, a fresh capture, and afterwards its own weights back.  The oracle (oracle/) is not involved."""
    import numpy as np
    from upflow_pytorch_amd import synthetic
    from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
    fixture = os.path.join('tests', 'golden', 'net_384x1280_hs1_robust.npz')
    z = np.load(os.path.join(ROOT, fixture))
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'net_meta.json')))['net_384x1280_hs1_robust']
    gf, gb = torch.from_numpy(z['flow_f_out']), torch.from_numpy(z['flow_b_out'])         # [2,2,384,1280] / [2,2,96,320] (every 4th pixel)
    ims = [synthetic.make_smooth_images(c, 1, 384, 1280) for c in (2, 12)]
    im1, im2 = torch.cat([a for a, _ in ims]).to(device), torch.cat([b for _, b in ims]).to(device)
    warps = [m for m in net.modules() if hasattr(m, 'mask_mode')]
    saved_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    saved_modes = [m.mask_mode for m in warps]
    try:
        net.load_state_dict(synthetic.make_state_dict(0, head_scale=1.0))     # (copy_ rounds every parameter from the fp32 values to ITS type)
        for m in warps:
            m.mask_mode = 'robust'
        n = max(1, int(streams)) if graph else 1
        idxs = [[0, 1, 1, 0] if s % 2 == 0 else [1, 0, 0, 1] for s in range(n)]
        outs = []
        if graph and n > 1:
            pipe = PipelinedInference(net, 4, 384, 1280, streams=n, device=device)
            for s, idx in enumerate(idxs):
                pipe.load(s, im1[idx].contiguous(), im2[idx].contiguous())
            for _ in range(3):
                for s in range(n):
                    pipe.replay(s)
            outs = [{k: v.float().cpu() for k, v in pipe.result(s).items()} for s in range(n)]
            del pipe
        elif graph:
            r = GraphedInference(net, 4, 384, 1280, device=device)
            r.load(im1[idxs[0]].contiguous(), im2[idxs[0]].contiguous())
            r.replay(); r.replay()
            torch.cuda.synchronize(device)
            outs = [{k: v.float().cpu() for k, v in r.out.items()}]
            del r
        else:
            with torch.no_grad():
                o = net({'im1': im1[idxs[0]].contiguous(), 'im2': im2[idxs[0]].contiguous(), 'if_loss': False})
            outs = [{k: v.float().cpu() for k, v in o.items()}]
        ef, eb, p99 = [], [], []
        for o, idx in zip(outs, idxs):
            per = (o['flow_f_out'] - gf[idx]).pow(2).sum(1).sqrt()
            ef.append(float(per.mean()))
            p99.append(float(per.flatten()[::7].quantile(0.99)))
            eb.append(float((o['flow_b_out'][:, :, ::4, ::4] - gb[idx]).pow(2).sum(1).sqrt().mean()))
            assert torch.isfinite(o['flow_f_out']).all()
        px = max(ef)
        return {'px': round(px, 5), 'pct_of_motion': round(100.0 * px / meta['mean_flow_px'], 4), 'px_backward': round(max(eb), 5), 'px_p99': round(max(p99), 4),
                'mean_flow_px': round(meta['mean_flow_px'], 3), 'fixture': fixture,
                'reference_self_sensitivity_px': meta['self_sensitivity_epe'],
                'path': '%s, 384x1280, batch 4, %s, %d step(s) in flight, every slot compared (worst reported); the network object the headline timed, '
                        "with the fixture's weights (full-scale heads) and exact-predicate warp mask" % (
                            'hipGraph' if graph else 'eager', 'x'.join(sorted({str(p.dtype).replace('torch.', '') for p in net.parameters()})), n),
                'definition': 'mean_pixels |flow - flow_reference|_2, forward flow (dataset/kitti_dataset.py:464-475 with mask = 1)'}
    finally:
        net.load_state_dict(saved_sd)
        for m, md in zip(warps, saved_modes):
            m.mask_mode = md


def train_roofline(net, B, H, W, ms_per_step):
    """The training step against the matrix-core peak: algorithmic flop of every convolution's forward, data gradient and weight
    gradient (3 x the forward's 2*k*k*Cin*Cout*pixels; the data gradient of the two layers that read the frames is not needed and
    not counted) / ms_per_step — the whole step incl. its memory-bound operators, losses and the optimizer, like roofline_step."""
    fwd, parts = conv_flop_per_step(net, B, H, W)
    first = 2.0 * 9 * 3 * 16 * 2 * B * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)          # pyramid stage 0, stride 2: no dgrad wrt the frames
    if net.sgi_model is not None:
        first += 2.0 * 9 * 3 * 16 * 2 * B * H * W                                         # SGU stem layer 0
    flop = 3.0 * fwd - first
    tf = flop / (ms_per_step * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tf / 2500.0, 4),
            'algorithmic_gflop_per_step': round(flop / 1e9, 1), 'forward_gflop': round(fwd / 1e9, 1)}


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: re-execute this command under torch.distributed.run with
    one rank per GPU of this node (RCCL over xGMI; rendezvous on 127.0.0.1, a free port), pass the ranks' output
    through and return their exit code.  Replaces the reference's single-process nn.DataParallel (utils/tools.py:130-148)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: the only mode the host driver supports
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def rank_spread(elapsed, steps, device=None):
    """{'min','max','slowest_rank'} of the ranks' own ms per step (one all_gather): a scaling run is diagnosable from its one line."""
    from upflow_pytorch_amd import parallel
    per = [e / steps * 1e3 for e in parallel.gather_over_ranks(elapsed, device)]
    return {'min': round(min(per), 3), 'max': round(max(per), 3), 'slowest_rank': per.index(max(per))}


def launch_check(args, rank, world):
    """--mode launch-check: the multi-rank plumbing of this file and nothing else — rendezvous, W warm-up + K timed "steps" (a
    rank-dependent sleep standing for the GPU work) bracketed by barriers exactly like the real modes, max-over-ranks, the per-rank
    spread, one JSON line from rank 0 — so that the 1/2/4/8-rank launch path is exercised on a box without GPUs
    (tests/test_distributed_cpu.py, --backend gloo)."""
    from upflow_pytorch_amd import parallel

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    step_s = 0.002 * (1 + rank % 3)                       # ranks deliberately differ: the line must report the slowest
    for _ in range(args.warmup):
        time.sleep(step_s)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(step_s)
    own = time.perf_counter() - t0                         # this rank's own steps (before it waits for the others)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    spread = rank_spread(own, args.steps)
    t = parallel.max_over_ranks(float(rank + 1))
    if rank == 0:
        print(json.dumps({'metric': 'launch-check', 'n_gpus': world, 'ranks': world, 'max_over_ranks': t, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'rank_ms_per_step': spread,
                          'backend': torch.distributed.get_backend() if world > 1 else None,
                          'master': '%s:%s' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')) if world > 1 else None}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--windows', type=int, default=3, help='timed windows of --steps steps each; value = the median window')
    ap.add_argument('--ramp-seconds', type=float, default=0.5, help='untimed replay before the warm-up steps (clock ramp)')
    ap.add_argument('--workload', default='config2', choices=sorted(WORKLOADS))
    ap.add_argument('--dtype', default=None, choices=sorted(DT))
    ap.add_argument('--pyramid-dtype', default=None, choices=['fp16', 'bf16'],
                    help='feature pyramid + 1x1 projections in this 16-bit type, everything downstream in --dtype (UPFlow_net.to_inference); '
                         'default: fp16 under --dtype bf16 (bf16 = every tensor bf16)')
    ap.add_argument('--batch', type=int, default=None, help='frame pairs per step and GPU (default: the workload\'s)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of a captured hipGraph')
    ap.add_argument('--streams', type=int, default=4,
                    help='steps in flight (inference): the captured step is replayed round-robin on this many HIP streams, each with its own '
                         'batch buffers (runtime.PipelinedInference); 1 = one step at a time')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eval-probe', action='store_true', help='skip `eval_batch1` (kitti_native: the batch-1 evaluation path, graph vs eager)')
    ap.add_argument('--no-epe-probe', action='store_true', help='skip `epe_vs_reference` (config 2: the timed path against the committed reference fixture)')
    ap.add_argument('--no-train-probe', action='store_true', help='skip the config-3 training step reported as `train_step`')
    ap.add_argument('--no-literal-split', action='store_true',
                    help="skip `literal_split` (the same step with the north star's literal split, feature pyramid through PyTorch-ROCm, timed beside the headline)")
    ap.add_argument('--torch-pyramid', action='store_true',
                    help="the north star's literal split: feature-pyramid convolutions through PyTorch-ROCm (MIOpen)")
    ap.add_argument('--fp32-conv', default='hip_x3', choices=['hip_x3', 'hip_x3s', 'miopen'],
                    help='--dtype fp32 (the parity mode): split-precision MFMA kernel (hip_x3s: low-order products in their own accumulators), or PyTorch-ROCm')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train', 'launch-check'],
                    help='train = BASELINE config 3: unsupervised step (fwd+loss+bwd+Adam), 256x832 crops, batch 4 per GPU, DDP')
    ap.add_argument('--backend', default=None, choices=['nccl', 'gloo'],
                    help='process-group backend (default nccl = RCCL; gloo only for --mode launch-check on a CPU box)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)                   # no external launcher: spawn one rank per GPU ourselves
    from upflow_pytorch_amd import parallel
    rank, world, local = parallel.init_from_env(backend=args.backend)
    assert world == max(args.gpus, 1), 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    if args.mode == 'launch-check':
        return launch_check(args, rank, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU path in the product)'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    if args.mode == 'train':
        return train_main(args, rank, world, device)
    B, H, W, dname = WORKLOADS[args.workload]
    B = args.batch or B
    dname = args.dtype or dname
    dtype = DT[dname]
    from upflow_pytorch_amd import synthetic as _weights
    # bf16: the feature pyramid + 1x1 projections (1.5 % of a step's flop, > 60 % of the bf16 path's distance to the reference) keep fp16
    # weights and features, everything downstream is bf16 — UPFlow_net.to_inference's default since round 6: 0.097 px instead of 0.179 px
    # to the reference (`epe_vs_reference`) for -0.9 % throughput (same box, alternated: 1855 / 1854 vs 1827 / 1850 pairs/s);
    # `--pyramid-dtype bf16` = every tensor bf16
    if args.pyramid_dtype is None and dtype == torch.bfloat16 and not args.torch_pyramid:
        args.pyramid_dtype = 'fp16'
    if args.pyramid_dtype == dname:
        args.pyramid_dtype = None
    net = build_net(dtype, device, hip_pyramid_convs=not args.torch_pyramid, fp32_conv=args.fp32_conv, pyramid_dtype=args.pyramid_dtype)
    im1, im2 = _weights.make_images(2 + rank, B, H, W)                      # every rank its own image pairs (the path shards by pair)
    im1, im2 = im1.to(device), im2.to(device)                               # inputs resident in HBM

    if args.no_graph:
        def step():
            with torch.no_grad():
                return net({'im1': im1, 'im2': im2, 'if_loss': False})
    elif args.streams > 1:
        # throughput mode: `streams` independent steps in flight (each the full forward on its own batch of B pairs, its own
        # static buffers), the coarse pyramid levels of one under the fine levels of another — runtime.PipelinedInference
        from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
        pipe = PipelinedInference(net, B, H, W, streams=args.streams, device=device)
        for slot in range(args.streams):
            a, b = _weights.make_images(2 + rank + 100 * slot, B, H, W)          # every slot its own image pairs
            pipe.load(slot, a.to(device), b.to(device))
        pipe.synchronize()
        last = [0]

        def step():
            last[0] = pipe.replay()
            return pipe.runners[last[0]].out
    else:
        from upflow_pytorch_amd.runtime import GraphedInference
        runner = GraphedInference(net, B, H, W, device=device)
        runner.load(im1, im2)
        step = runner.replay

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # clock ramp: the part idles at a low clock; a W of 10 steps is 35 ms of work, shorter than the governor's ramp, so
    # the same (untimed, un-counted) step is replayed for --ramp-seconds before the W warm-up steps the contract asks for
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        step()
        torch.cuda.synchronize(device)
    for _ in range(args.warmup):
        step()
    # `--windows` timed windows of EXACTLY K steps each, every one bracketed by barrier + synchronize on both sides and reduced with MAX
    # over ranks; `value` is the MEDIAN window, value_min / value_max the others (one 45 ms sample said nothing about repeatability)
    windows = []
    for _w in range(max(1, args.windows)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize(device)
        own = time.perf_counter() - t0                     # this rank's own steps, before it waits for the others (rank_ms_per_step)
        barrier()
        windows.append((parallel.max_over_ranks(time.perf_counter() - t0, device), own))
    windows.sort()
    elapsed, own = windows[len(windows) // 2]
    spread = rank_spread(own, args.steps, device)
    assert torch.isfinite(out['flow_f_out']).all()
    pipelined = (not args.no_graph) and args.streams > 1
    single = None
    if pipelined:
        for r_ in pipe.runners:                                  # every slot's outputs, not only the last step's
            assert torch.isfinite(r_.out['flow_f_out']).all() and torch.isfinite(r_.out['flow_b_out']).all()
        # the same K steps with ONE step in flight (the latency-bound schedule every earlier round reported), for continuity
        r0 = pipe.runners[0]
        for _ in range(args.warmup):
            r0.replay()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            r0.replay()
        barrier()
        el1 = parallel.max_over_ranks(time.perf_counter() - t1, device)
        single = {'value': round(world * B * args.steps / el1, 3), 'unit': 'frame-pairs/s', 'ms_per_step': round(el1 / args.steps * 1e3, 3),
                  'note': 'one step in flight: the per-step latency; the headline keeps %d independent steps in flight on %d HIP streams' % (args.streams, args.streams)}

    if rank == 0:
        pairs = world * B * args.steps
        line = {
            'metric': 'frame-pairs/sec at 384x1280 bf16' if args.workload == 'config2' else 'frame-pairs/sec',
            'value': round(pairs / elapsed, 3), 'value_min': round(pairs / windows[-1][0], 3), 'value_max': round(pairs / windows[0][0], 3),
            'windows': len(windows), 'unit': 'frame-pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dname, 'data': 'synthetic',
            'config': {'workload': '%s: UPFlow_net inference forward (flow fwd+bwd, occlusion masks, SGU on), '
                                   '%dx%d, batch %d per GPU, random-init weights' % (args.workload, H, W, B),
                       'global_batch': world * B, 'parallelism': 'replicas x%d (image pairs sharded, no collective)' % world,
                       'ranks': world, 'backend': (torch.distributed.get_backend() + ' (RCCL)') if world > 1 else None,
                       'rank_ms_per_step': spread,
                       'hip_graph': not args.no_graph, 'capture_fallback': False,
                       'steps_in_flight': args.streams if pipelined else 1,
                       'pyramid_convs': 'PyTorch-ROCm' if args.torch_pyramid or (dtype == torch.float32 and args.fp32_conv == 'miopen') else 'HIP (MFMA kernel)',
                       'fp32_conv': args.fp32_conv if dtype == torch.float32 else None,
                       'pyramid_dtype': args.pyramid_dtype},
            'roofline': roofline_probe(B, H, W, dtype, device, feature_dtype=DT[args.pyramid_dtype] if args.pyramid_dtype else None),
        }
        conv_rf = conv_roofline_probe(B, H, W, dtype, device, args.fp32_conv)
        if conv_rf is not None:
            line['roofline_conv'] = conv_rf
        if dtype != torch.float32 or args.fp32_conv != 'miopen':
            # the whole step against the matrix-core peak: every convolution's algorithmic flop / the
            # step time bench.py reports — the efficiency of the STEP, not of its best layer (`roofline_conv`); in the fp32 parity
            # mode the split-precision kernel spends three matrix products per operand pair, so its ceiling is a third of the peak
            flop, parts = conv_flop_per_step(net, B, H, W)
            tf = flop / (elapsed / args.steps) / 1e12
            line['roofline_step'] = {'bound': 'mfma', 'achieved': round(tf, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(tf / 2500.0, 4),
                                     'algorithmic_gflop_per_step': round(flop / 1e9, 1),
                                     'gflop_by_stack': {k: round(v / 1e9, 1) for k, v in parts.items()},
                                     'note': 'sum of 2*k*k*Cin*Cout*pixels over the convolutions of one step / ms_per_step; includes the memory-bound operators\' time'}
        if world == 1 and not args.no_literal_split and not args.torch_pyramid and dtype != torch.float32 and not args.no_graph:
            # north_star's letter ("the feature-pyramid convolutions stay PyTorch-ROCm"): the same workload with the pyramid
            # through MIOpen, timed in this run beside the headline (which runs the pyramid on the hand-written kernel too)
            try:
                net_ls = build_net(dtype, device, hip_pyramid_convs=False)
                r2 = GraphedInference(net_ls, B, H, W, device=device)
                r2.load(im1, im2)
                for _ in range(10):
                    r2.replay()
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                for _ in range(20):
                    r2.replay()
                torch.cuda.synchronize(device)
                ms = (time.perf_counter() - t1) / 20 * 1e3
                line['literal_split'] = {'pyramid_convs': 'PyTorch-ROCm (MIOpen)', 'value': round(B * 1e3 / ms, 2), 'unit': 'frame-pairs/s',
                                         'ms_per_step': round(ms, 3), 'steps': 20, 'steps_in_flight': 1}
                del r2
                if args.streams > 1:                          # ... and with as many steps in flight as the headline
                    p2 = PipelinedInference(net_ls, B, H, W, streams=args.streams, device=device)
                    for s_ in range(args.streams):
                        p2.load(s_, im1, im2)
                    for _ in range(3 * args.streams):
                        p2.replay()
                    p2.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(40):
                        p2.replay()
                    p2.synchronize()
                    ms = (time.perf_counter() - t1) / 40 * 1e3
                    line['literal_split']['pipelined'] = {'value': round(B * 1e3 / ms, 2), 'ms_per_step': round(ms, 3), 'steps': 40,
                                                          'steps_in_flight': args.streams}
                    del p2
                del net_ls
            except Exception as e:                            # (an extra: it must never take the headline line down)
                line['literal_split'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if single is not None:
            line['one_step_in_flight'] = single
        if world == 1 and not args.no_epe_probe and (H, W) == (384, 1280):
            try:
                line['epe_vs_reference'] = epe_probe(net, args.streams, not args.no_graph, device)
            except Exception as e:                            # (must never take the headline line down)
                line['epe_vs_reference'] = {'error': '%s: %s' % (type(e).__name__, e)}
        if world == 1 and args.workload == 'kitti_native' and not args.no_eval_probe:
            line['eval_batch1'] = eval_probe(net, H, W, device)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        if world == 1 and not args.no_train_probe and args.workload == 'config2':
            line['train_step'] = train_probe(device)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
