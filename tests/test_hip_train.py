"""Train-mode parity (BASELINE config 3 at a small crop): loss terms and per-parameter gradient norms of
the HIP path against the reference's (tests/golden/train_128x192.npz), robust mask on both sides."""
import numpy as np
import pytest
import torch

import oracle
import _weights
from conftest import load_golden

pytestmark = pytest.mark.gpu
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def build():
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    return net.cuda().train()


def test_train_losses_and_gradients_match_reference():
    g = load_golden('train_128x192')
    net = build()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    batch['if_loss'] = True
    out = net(batch)
    terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
    for k, v in terms.items():
        want = float(g[k])
        print(k, float(v), want)
        assert abs(float(v) - want) <= 2e-4 * max(1.0, abs(want)), k
    assert oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out']) <= 1e-4
    sum(terms.values()).backward()
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    print('max rel grad-norm error %.3g (param %s)' % (rel.max(), names[int(rel.argmax())]))
    assert (got > 0).all(), 'every parameter must receive a gradient'
    assert rel.max() <= 2e-3
    # direction, not only size: cosine of the 64-projection fingerprint of every gradient with the reference's, and every
    # bias gradient element by element
    cos, worst = grad_direction_check({n: params[n].grad for n in names}, g)
    print('gradient direction: min cosine %.7f (param %s), worst bias-gradient error %.3g' % (cos.min(), names[int(cos.argmin())], worst))
    assert cos.min() >= 0.9999 and worst <= 5e-3


def grad_direction_check(named_grads, g):
    """-> (cosine of the projection fingerprints per parameter, worst relative error of a bias gradient)."""
    names = sorted(named_grads)
    got, want = _weights.grad_projections(named_grads), g['grad_proj'].numpy()
    cos = (got * want).sum(1) / np.maximum(np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1), 1e-30)
    worst = 0.0
    for i, n in enumerate(names):
        if n.endswith('.bias'):
            w = g['gbias_%d' % i].double()
            worst = max(worst, float((named_grads[n].detach().double().cpu() - w).norm() / w.norm().clamp_min(1e-12)))
    return cos, worst


def test_trainer_step_updates_every_parameter():
    """Trainer.step = forward + loss + backward + Adam(amsgrad) on the HIP path: finite loss terms, every
    parameter moves.  (No "loss goes down" claim: with random MSRA weights the self-distillation term makes
    the first steps of the reference's own recipe noisy.)"""
    from upflow_pytorch_amd.train import Trainer
    net = build()
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    tr = Trainer(net, lr=1e-4, distributed=False)
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    stats = tr.step(batch)
    assert set(stats) == {'loss', 'photo_loss', 'smooth_loss', 'census_loss', 'msd_loss'}
    assert all(np.isfinite(v) for v in stats.values())
    moved = [n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert len(moved) == 80


def test_trainer_under_rccl_ddp_real_net():
    """Row (e): the REAL UPFlow_net (ctypes autograd Functions, stacked training schedule) wrapped in
    DistributedDataParallel on an `nccl` (= RCCL) process group — world size 1 on this 1-GPU box: the DDP reducer, the
    25 MB bucket with gradient_as_bucket_view and the RCCL all-reduce run for real; with one rank the averaged gradient
    must equal the plain single-process gradient, and the Trainer's loss all-reduce must return the same terms.
    Runs in its OWN process (tests/_ddp_one_rank.py): a process group leaves RCCL's watchdog / heartbeat threads behind in
    the process that created it, and those threads poll HIP events — an error for every later hipGraph capture of the pytest
    process under the default (global) capture mode."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, '_ddp_one_rank.py')], capture_output=True, text=True, timeout=900)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stderr[-4000:]
    assert 'DDP-ONE-RANK-OK' in out.stdout


@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
def test_graphed_training_step_equals_eager(mode):
    """Trainer(graph=True): after 3 eager steps the whole step (forward, losses, backward, Adam) is ONE captured hipGraph.
    Five steps graphed vs five steps eager from the same weights on the same batch: same loss terms, same parameters
    (to the run-to-run reproducibility of MIOpen's gradient kernels, which use atomics)."""
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.train import Trainer

    def make():
        conf = UPFlow_net.config()
        d = dict(FLAGS)
        d.update(_weights.TRAIN_FLAGS)
        d['train_conv_dtype'] = mode
        conf.update(d, verbose=False)
        net = conf()
        net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
        return net
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    res = []
    for graph in (False, False, True):
        tr = Trainer(make(), lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=graph)
        stats = [tr.step(batch) for _ in range(5)]
        assert (tr._graph is not None) == graph
        res.append((stats, {n: p.detach().clone() for n, p in tr.raw_net.named_parameters()}))
    (s0, p0), (_, pe), (s1, p1) = res
    for a, b in zip(s0, s1):
        for k in a:
            # (1e-2, not bit for bit: capturable Adam is another floating-point spelling of the update, and at this 128x192 size
            # the two coarsest levels take PyTorch-ROCm's convolution gradients, which are not reproducible run to run)
            assert abs(a[k] - b[k]) <= 1e-2 * max(1.0, abs(a[k])), (k, a[k], b[k])
    diff = lambda u, v: (max(float((u[n] - v[n]).abs().max()) for n in u),
                         sum(float((u[n] - v[n]).abs().sum()) for n in u) / sum(u[n].numel() for n in u))
    (worst, mean), (noise_worst, noise_mean) = diff(p0, p1), diff(p0, pe)
    print('graphed vs eager after 5 steps (%s): parameter difference worst %.3g mean %.3g; eager vs eager worst %.3g mean %.3g'
          % (mode, worst, mean, noise_worst, noise_mean))
    # lr 1e-4, 5 Adam steps: a parameter moves by <= 5e-4, two runs differ by <= 1e-3; the stride-2 layers' gradients come
    # from MIOpen kernels with atomics, and Adam's first steps turn a sign flip of a near-zero gradient into +-lr — so the
    # yardstick is the difference between two EAGER runs
    assert worst <= 1e-3 and mean <= max(3.0 * noise_mean, 2e-5), (worst, mean, noise_worst, noise_mean)


def test_bf16_training_mode_tracks_the_fp32_reference():
    """train_conv_dtype='bf16' (decoder + pyramid activations in bf16, fp32 master weights, forward / dgrad / wgrad on the
    matrix cores) against the REFERENCE's fp32 losses and gradient norms (tests/golden/train_128x192.npz): new behaviour
    (the reference trains in fp32 only), so the bound is the bf16 rounding envelope, not 2e-4: losses within 0.2 %, every
    parameter's gradient norm within 4 %, its direction no further from the reference's gradient than the reference's own
    gradient moves when its input frames are rounded to bf16, the flow within 0.02 px."""
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    g = load_golden('train_128x192')
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    d['train_conv_dtype'] = 'bf16'
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    net = net.cuda().train()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    batch['if_loss'] = True
    out = net(batch)
    terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
    for k, v in terms.items():
        assert abs(float(v) - float(g[k])) <= 2e-3 * max(1.0, abs(float(g[k]))), (k, float(v), float(g[k]))
    assert oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out']) <= 0.02
    sum(terms.values()).backward()
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    print('bf16 training mode: max rel grad-norm error %.3g (param %s), median %.3g' % (rel.max(), names[int(rel.argmax())], np.median(rel)))
    assert all(params[n].grad.dtype == torch.float32 for n in names) and rel.max() <= 4e-2
    # direction: the model's hard masks make its gradients discontinuous, so the yardstick is the REFERENCE's own sensitivity
    # to a 16-bit-sized perturbation — its fp32 step with nothing but the input frames rounded to bf16 (golden
    # grad_proj_bf16_frames: cosine to its exact gradients min 0.987, median 0.9987; fp16 activations land on the same floor)
    cos, worst = grad_direction_check({n: params[n].grad for n in names}, g)
    ref, r16 = g['grad_proj'].numpy(), g['grad_proj_bf16_frames'].numpy()
    floor = (ref * r16).sum(1) / (np.linalg.norm(ref, axis=1) * np.linalg.norm(r16, axis=1))
    print('bf16 training mode: gradient cosine min %.5f (param %s) median %.5f | reference with bf16-rounded frames: min %.5f median %.5f'
          % (cos.min(), names[int(cos.argmin())], np.median(cos), floor.min(), np.median(floor)))
    assert np.median(cos) >= np.median(floor) - 1e-3 and cos.min() >= floor.min() - 0.01 and (cos >= floor - 0.02).all()


@pytest.mark.parametrize('variant', ['no_sgu', 'per_direction', 'no_sinks', 'layerwise_stacks', 'fp16', 'frozen_pyramid'])
def test_bf16_training_schedule_variants_agree(variant):
    """The matrix-core training mode under the schedule / flag variants a user can select — without the self-guided upsampling,
    the reference's per-direction schedule (stacked_training = False), without the parameter sinks, with the layer-wise dense
    stacks, in fp16, with a frozen feature pyramid: every one gives the default schedule's losses (same arithmetic up to the
    summation order of gradient accumulations) and finite gradients for every trainable parameter."""
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.model.pwc_modules import _DenseStack

    def run(var):
        conf = UPFlow_net.config()
        d = dict(FLAGS)
        d.update(_weights.TRAIN_FLAGS)
        d['train_conv_dtype'] = 'fp16' if var == 'fp16' else 'bf16'
        if var == 'no_sgu':
            d['if_sgu_upsample'] = False
        conf.update(d, verbose=False)
        net = conf()
        sd = _weights.make_state_dict(0, head_scale=0.1)
        net.load_state_dict(sd, strict=(var != 'no_sgu'))
        net = net.cuda().train()
        if var == 'per_direction':
            net.stacked_training = False
        if var == 'no_sinks':
            net.shared_grad_sinks = False
        if var == 'layerwise_stacks':
            for m in net.modules():
                if isinstance(m, _DenseStack):
                    m._no_train_buffer = True
        if var == 'frozen_pyramid':
            for p in net.feature_pyramid_extractor.parameters():
                p.requires_grad_(False)
        batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
        batch['if_loss'] = True
        out = net(batch)
        terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
        sum(terms.values()).backward()
        for n, p in net.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
            else:
                assert p.grad is None, n
        gn = float(torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None]).norm())
        return {k: float(v) for k, v in terms.items()}, gn
    got, gn = run(variant)
    if variant in ('no_sgu', 'fp16'):
        ref, gn_ref = run(variant)                      # (different arithmetic: only reproducibility is checked)
        tol = 1e-3
    else:
        ref, gn_ref = run('default')
        tol = 5e-3                                     # bf16 intermediates summed in a different order
    for k in ref:
        assert abs(got[k] - ref[k]) <= tol * max(1.0, abs(ref[k])), (k, got[k], ref[k])
    if variant != 'frozen_pyramid':
        assert abs(gn - gn_ref) <= 0.05 * gn_ref, (gn, gn_ref)


def _config3_trainer(mode, graph, B=4, seed=0):
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.train import Trainer
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    d['train_conv_dtype'] = mode
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(seed, head_scale=0.1))
    return Trainer(net, lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=graph)


def test_config3_full_size_step_graphed_equals_eager_and_bf16_tracks_fp32():
    """BASELINE config 3 at its real size (256x832 crops of 288x864 frames, batch 4 per GPU,
    /root/reference/dataset/kitti_dataset.py:268-342): six steps eager and six steps with the step captured as one hipGraph give
    the same loss trajectory, and the bf16 matrix-core mode stays inside 0.5 % of the fp32 mode's loss terms at the first step (same
    weights) and 3 % over the following five (two trajectories of a model with hard masks)."""
    from upflow_pytorch_amd.train import synthetic_train_batch
    batch = synthetic_train_batch(4, device='cuda')
    traj = {}
    for mode, graph in (('bf16', False), ('bf16', True), ('fp32', False)):
        tr = _config3_trainer(mode, graph)
        traj[(mode, graph)] = [tr.step(batch) for _ in range(6)]
        assert (tr._graph is not None) == graph
        del tr
        torch.cuda.empty_cache()
    for a, b in zip(traj[('bf16', False)], traj[('bf16', True)]):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * max(1.0, abs(a[k])), ('graph vs eager', k, a[k], b[k])
    for i, (a, b) in enumerate(zip(traj[('bf16', False)], traj[('fp32', False)])):
        print('step %d  bf16 %s | fp32 %s' % (i, {k: round(v, 4) for k, v in a.items()}, {k: round(v, 4) for k, v in b.items()}))
        for k in a:       # same weights at step 0: the 16-bit rounding envelope; afterwards two trajectories that drift apart
            assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= (5e-3 if i == 0 else 3e-2) * max(1.0, abs(b[k])), ('bf16 vs fp32', i, k, a[k], b[k])


def test_config3_full_size_bf16_training_is_bit_reproducible():
    """At config 3's real size every level of the pyramid takes the hand-written kernels (the coarsest is 4x13), whose scatters
    accumulate in fixed point and whose split-K sums run in a fixed order: two fresh runs of six eager steps end with BIT-IDENTICAL
    parameters and loss terms, and so do two runs with the step captured as a hipGraph.  (At the 128x192 size of the other tests
    the two coarsest levels are narrower than 8 pixels and fall back to PyTorch-ROCm's convolution gradients, which are not
    reproducible run to run — tools/det_probe*.py; those tests carry tolerances for that.)"""
    from upflow_pytorch_amd.train import synthetic_train_batch
    batch = synthetic_train_batch(4, device='cuda')
    ends = []
    for graph in (False, False, True, True):
        tr = _config3_trainer('bf16', graph)
        stats = [tr.step(batch) for _ in range(6)]
        assert (tr._graph is not None) == graph, getattr(tr, 'capture_error', None)
        ends.append((stats, torch.cat([p.detach().flatten().clone() for p in tr.raw_net.parameters()])))
        del tr
        torch.cuda.empty_cache()
    # (eager vs graphed is NOT bit for bit: the graphed trainer's Adam is the capturable form with the learning rate and the
    # step count on the device — the same update in another floating-point spelling; test_config3_full_size_step_... bounds it)
    for name, i, j in (('two eager runs', 0, 1), ('two graphed runs', 2, 3)):
        assert ends[i][0] == ends[j][0], (name, ends[i][0][-1], ends[j][0][-1])
        assert torch.equal(ends[i][1], ends[j][1]), name


def _traj_trainer(mode, graph, distill):
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.train import Trainer
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    d['train_conv_dtype'] = mode
    if not distill:
        d['multi_scale_distillation_weight'] = 0          # (the reference's default, model/upflow.py:312)
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    return Trainer(net, lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=graph)


@pytest.mark.parametrize('mode,graph', [('fp32', False), ('fp32', True), ('bf16', True)])
def test_training_learns_a_known_motion_like_the_reference(mode, graph):
    """Training must REDUCE something, not only move parameters.  One synthetic pair whose true motion is a
    2-pixel horizontal shift; the unsupervised recipe of the reference (photometric + smoothness + census,
    model/upflow.py:394-491; Adam(amsgrad) lr 1e-4 wd 1e-4, scripts/simple_train.py:121-122; pyramid distillation at its
    default weight 0).  200 steps: the loss falls to < 0.3x, the photometric term to < 0.55x, the end-point error against
    the KNOWN motion below 0.15 px — and the loss terms follow the trajectory the REFERENCE itself takes on this batch (asserted: every check point within 30 %, the plateau of steps 60-120 within 5 % — 10 % for the small smoothness term; nine runs measured 0.1 - 1.5 % on the plateau, up to 10 % at single check points)
    (tests/golden/train_traj_128x192.json, generated by make_golden.py `traj` from the imported reference on CPU).
    This test is what exposed the hipMemsetAsync-in-hipGraph ordering fault (csrc/common.hpp: zero_fill_u64): before that
    fix the graphed step diverged after a timing-dependent number of replays while the eager step was fine."""
    import json
    import os
    from conftest import GOLDEN
    from upflow_pytorch_amd.train import synthetic_train_batch
    ref = {r['step']: r for r in json.load(open(os.path.join(GOLDEN, 'train_traj_128x192.json')))['no_distillation']}
    batch = synthetic_train_batch(2, crop_hw=(128, 192), raw_hw=(160, 256), device='cuda')
    tr = _traj_trainer(mode, graph, distill=False)
    gt = torch.zeros(2, 2, 128, 192, device='cuda')
    gt[:, 0] = 2.0                                          # im2(x) = im1(x - 2): the forward flow is (+2, 0)
    first = last = None
    seen = {}
    for i in range(201):
        s = tr.step(batch, sync_stats=(i % 20 == 0))
        if i % 20 == 0:
            first = first or s
            last = s
            if i in ref:                                    # the reference's trajectory (steps 0 .. 120)
                seen[i] = s
    # On the reference's curve.  Adam on one pair is a chaotic trajectory and MIOpen's fp32 gradient kernels / our atomics sum
    # in arrival order: runs of the SAME build differ by up to 10 % at a single check point (measured), and on the steep part
    # (steps 20-40: the census term falls 3.8 % per step) a shift of two steps is 8 %.  So: step 0 exactly; every check point in
    # the right place (30 %); and the plateau the run settles on — the mean over steps 60 .. 120 — within 5 % of the reference's (10 % for smooth_loss, whose plateau value is ~0.02; nine runs: 0.1 - 1.5 %).
    keys = ('photo_loss', 'smooth_loss', 'census_loss')
    for k in keys:
        assert abs(seen[0][k] - ref[0][k]) <= (2e-4 if mode == 'fp32' else 2e-3) * max(abs(ref[0][k]), 0.05), (0, k, seen[0][k], ref[0][k])
        dev = {i: abs(seen[i][k] - ref[i][k]) / max(abs(ref[i][k]), 0.05) for i in seen}
        print('%s: largest single-point deviation %.1f %% (step %d)' % (k, 100 * max(dev.values()), max(dev, key=dev.get)))
        for i in seen:
            assert dev[i] <= 0.30, (i, k, seen[i][k], ref[i][k])
        late = [i for i in seen if i >= 60]
        ours, theirs = sum(seen[i][k] for i in late) / len(late), sum(ref[i][k] for i in late) / len(late)
        print('%s: plateau (steps 60-120) %.4f, reference %.4f' % (k, ours, theirs))
        assert abs(ours - theirs) <= (0.05 if k != 'smooth_loss' else 0.10) * max(abs(theirs), 0.05), (k, ours, theirs)
    tr.raw_net.eval()
    with torch.no_grad():
        f = tr.raw_net(dict(batch, if_loss=False))['flow_f_out'].float()[:, :, 16:-16, 16:-16]
    epe = float((f - gt[:, :, 16:-16, 16:-16]).pow(2).sum(1).sqrt().mean())
    print('%s %s: loss %.4f -> %.4f, photo %.4f -> %.4f, census %.4f -> %.4f, EPE vs the known 2-px motion %.3f px (mean u %.3f)'
          % (mode, 'graph' if graph else 'eager', first['loss'], last['loss'], first['photo_loss'], last['photo_loss'],
             first['census_loss'], last['census_loss'], epe, float(f[:, 0].mean())))
    assert (tr._graph is not None) == graph, 'requested graph=%s, capture_fallback=%s (%s)' % (graph, tr.capture_fallback, getattr(tr, 'capture_error', None))
    assert last['loss'] <= 0.3 * first['loss'] and last['photo_loss'] <= 0.55 * first['photo_loss'] and epe <= 0.15


def test_training_trajectory_with_distillation_follows_the_reference():
    """The FULL loss (with the pyramid-distillation term, weight 1) on the same batch: chaotic — the reference's own run leaves
    a loss of 10.6 for 37 at step 20 and comes back — and the fp32 HIP path goes through the same excursion: loss terms within
    8 % of the reference's at steps 0, 20 ... 100."""
    import json
    import os
    from conftest import GOLDEN
    from upflow_pytorch_amd.train import synthetic_train_batch
    ref = {r['step']: r for r in json.load(open(os.path.join(GOLDEN, 'train_traj_128x192.json')))['full']}
    batch = synthetic_train_batch(2, crop_hw=(128, 192), raw_hw=(160, 256), device='cuda')
    tr = _traj_trainer('fp32', True, distill=True)
    for i in range(101):
        s = tr.step(batch, sync_stats=(i % 20 == 0))
        if i % 20 == 0:
            print('step %3d  build %s | reference %s' % (i, {k: round(v, 4) for k, v in s.items()}, {k: round(v, 4) for k, v in ref[i].items() if k in s}))
            for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss'):
                assert abs(s[k] - ref[i][k]) <= (2e-4 if i == 0 else 0.08) * max(abs(ref[i][k]), 0.05), (i, k, s[k], ref[i][k])


def test_graph_mode_follows_the_lr_scheduler():
    """capturable Adam bakes a python-float lr into the captured graph, so ExponentialLR was silently ignored in
    graph mode.  The lr is a device tensor now: after end_epoch() with gamma 0.1 a replayed step moves the parameters ~10x less."""
    from upflow_pytorch_amd.train import Trainer
    net = build()
    tr = Trainer(net, lr=1e-3, scheduler_gamma=0.1, device=torch.device('cuda', 0), distributed=False, graph=True)
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    for _ in range(tr.graph_warmup + 1):
        tr.step(batch)
    assert tr._graph is not None

    def delta():
        before = torch.cat([p.detach().flatten().clone() for p in tr.raw_net.parameters()])
        tr.step(batch)
        return float((torch.cat([p.detach().flatten() for p in tr.raw_net.parameters()]) - before).abs().mean())
    d0 = delta()
    tr.end_epoch()
    d1 = delta()
    print('mean parameter step %.3g -> %.3g after one ExponentialLR(0.1) epoch' % (d0, d1))
    assert d1 <= 0.2 * d0 and d1 >= 0.05 * d0


def test_capture_failure_leaves_no_poisoned_caches(monkeypatch):
    """a failed capture had put packed-weight cache entries into graph-pool memory that was never written.  The
    caches are dropped after every capture attempt now: steps after a forced failure equal pure eager steps."""
    from upflow_pytorch_amd import train as train_mod
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    ref = _config3_trainer('bf16', False)
    want = [ref.step(batch) for _ in range(6)]
    tr = _config3_trainer('bf16', True)
    real = tr._step_body
    calls = {'n': 0}

    def failing(b):
        calls['n'] += 1
        out = real(b)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('forced capture failure')
        return out
    monkeypatch.setattr(tr, '_step_body', failing)
    with pytest.warns(UserWarning):
        got = [tr.step(batch) for _ in range(6)]
    assert tr._graph is None and tr.capture_fallback and not tr.use_graph
    for a, b in zip(got, want):
        for k in a:
            # (1e-2: at this 128x192 size the two coarsest levels are < 8 pixels wide and take PyTorch-ROCm's convolution
            # gradients, which differ from run to run by ~1e-3; poisoned weights would show as garbage or NaN)
            assert abs(a[k] - b[k]) <= 1e-2 * max(1.0, abs(b[k])), (k, a[k], b[k])


def test_replay_rejects_a_different_batch_shape():
    from upflow_pytorch_amd.train import Trainer
    tr = Trainer(build(), lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=True)
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    for _ in range(tr.graph_warmup + 1):
        tr.step(batch)
    assert tr._graph is not None
    small = {k: v[:1] for k, v in batch.items()}
    with pytest.warns(UserWarning):
        s = tr.step(small)                                  # a last partial batch: one eager step, the graph stays
    assert all(np.isfinite(v) for v in s.values()) and tr._graph is not None
    assert all(np.isfinite(v) for v in tr.step(batch).values())


@pytest.mark.parametrize('frozen', [False, True, 'fused', 'bumped'])
def test_captured_graph_survives_cache_clears_allocator_churn_and_an_eager_step(frozen):
    """ADVICE r3 (high).  The captured step reads, at addresses baked into the graph, tensors that were allocated BEFORE the
    capture: the zero-bias operand of the data-gradient convolutions, the packs of parameters whose version did not move
    (frozen ones), the loss module's constants.  Trainer._capture used to drop the only references to them; the next eager
    allocation could land on that memory and the replays then read it.  Config 3's full size (every level on the deterministic
    kernels: two graphed runs are bit-identical, test_config3_full_size_bf16_training_is_bit_reproducible), two runs of
    [capture, 2 replays, one eager step on a partial batch, 3 replays]; the second run also clears every cache, returns the
    cached blocks to the allocator and fills fresh NaN tensors of the sizes in question after the capture and after the eager
    step.  Both runs must end with bit-identical parameters and loss terms."""
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.train import synthetic_train_batch
    from upflow_pytorch_amd.utils import loss as loss_mod
    batch = synthetic_train_batch(4, device='cuda')
    small = {k: v[:1].contiguous() for k, v in batch.items()}

    def churn(hold):
        ops.train_caches_clear()
        loss_mod._VALID.clear()
        loss_mod._ZEROS.clear()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        for nbytes in (512, 4096, 16384, 65536, 1 << 20, 4 << 20):
            for _ in range(24):
                hold.append(torch.full((nbytes // 4,), float('nan'), dtype=torch.float32, device='cuda'))
        torch.cuda.synchronize()

    ends = []
    for do_churn in (False, True):
        tr = _config3_trainer('bf16', True)
        if frozen:                                            # frozen pyramid / decoder: their packs are cache hits inside the capture
            tr.raw_net.froze_PWC()
            # True: the multi-launch (foreach) optimizer; 'fused': the one-kernel optimizer the Trainer uses by default, which does
            # not advance the parameters' version counters itself (the Trainer's `optimizer` setter registers ops.register_version_hook on
            # whatever optimizer it is given: the packed weight copies are keyed on those counters)
            tr.optimizer = torch.optim.Adam([p for p in tr.net.parameters() if p.requires_grad], lr=tr.optimizer.param_groups[0]['lr'],
                                            amsgrad=True, weight_decay=1e-4, capturable=True, fused=(frozen == 'fused'))
        if frozen == 'bumped':
            # round 4: every parameter's version — the FROZEN ones' too — moves after every step body, so nothing is a cache hit
            # inside the capture and the capture's allocation pattern shifts.  That run used to return census_loss = 9.8e3 (the
            # true value: 2.0) from the replays after the NaN fill: not a lifetime fault of ours (no pre-capture block is freed
            # while the graph lives: tools/frozen_graph_probe.py HISTORY=1) but ATen's multi-block mean() inside the captured
            # step, which zeroes its semaphores with a memset node; the census term now reduces through csrc/loss.hip.
            body = tr._step_body

            def bumped(b, body=body, tr=tr):
                r = body(b)
                torch.autograd.graph.increment_version(list(tr.net.parameters()))
                return r
            tr._step_body = bumped
        hold, stats = [], []
        for _ in range(tr.graph_warmup + 1):
            stats.append(tr.step(batch))
        assert tr._graph is not None and tr._graph_keepalive, tr.capture_error
        if do_churn:
            churn(hold)
        stats += [tr.step(batch) for _ in range(2)]
        with pytest.warns(UserWarning):
            stats.append(tr.step(small))                      # a last partial batch: one eager step, new allocations
        if do_churn:
            churn(hold)
        stats += [tr.step(batch) for _ in range(3)]
        assert all(np.isfinite(v) for s in stats for v in s.values())
        ends.append((stats, torch.cat([p.detach().flatten().clone() for p in tr.raw_net.parameters()])))
        del tr, hold
        torch.cuda.empty_cache()
    assert ends[0][0] == ends[1][0], (ends[0][0][-1], ends[1][0][-1])
    assert torch.equal(ends[0][1], ends[1][1])


def test_validation_between_replayed_steps_sees_the_current_weights():
    """ADVICE r4 (medium).  A REPLAYED training step changes the parameters on the device without running any host code, so
    nothing advanced their autograd version counters — and a validation forward (net.eval() under no_grad) between training
    steps goes through pwc_modules._PackedConv*, whose packed weight copies are keyed on those counters: eval, replay, eval
    multiplied by the weights of the FIRST eval.  Trainer.step advances the versions after every replay now.  Protocol: graph-
    train, eval, replay 5 more steps, eval again — the second eval must equal an eval after invalidate_packed() bit for bit,
    differ from the first, and GraphedInference.check_weights must notice the change too."""
    from upflow_pytorch_amd.train import Trainer
    from upflow_pytorch_amd.runtime import GraphedInference
    net = build()
    tr = Trainer(net, lr=1e-3, device=torch.device('cuda', 0), distributed=False, graph=True)
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    for _ in range(tr.graph_warmup + 1):
        tr.step(batch)
    assert tr._graph is not None
    im1, im2 = batch['im1'], batch['im2']

    def validate():
        tr.raw_net.eval()
        with torch.no_grad():        # (fp32 inference: every convolution through _PackedConv3x3 -> the split-precision kernel)
            return tr.raw_net({'im1': im1, 'im2': im2, 'if_loss': False})['flow_f_out'].clone()
    first = validate()
    runner = GraphedInference(tr.raw_net.eval(), im1.shape[0], im1.shape[2], im1.shape[3], device=im1.device)
    for _ in range(5):
        tr.step(batch)                                   # replays
    second = validate()
    tr.raw_net.invalidate_packed()
    fresh = validate()
    assert torch.equal(second, fresh), 'a validation forward after replayed steps used stale packed weights'
    assert not torch.equal(first, second), 'five optimizer steps at lr 1e-3 must move the validation output'
    with pytest.raises(RuntimeError):
        runner.replay()                                  # the captured inference graph reads the packed copies made before


def _realistic_step(train_conv_dtype):
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    d['train_conv_dtype'] = train_conv_dtype
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=1.0))
    net = net.cuda().train()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch(**_weights.TRAIN_HS1).items()}
    batch['if_loss'] = True
    out = net(batch)
    terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
    sum(terms.values()).backward()
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    return out, terms, names, params


def test_train_step_at_realistic_motion_matches_reference():
    """training parity where the flows LEAVE the crop (tests/golden/train_128x416_hs1.npz: full-scale heads, mean
    |flow| 11.4 px, crop 3 / 2 px from the frame's corner: 5.5 % of the photometric samples outside the crop, 1 % clamped at the frame
    border, 94 % of the pixels failing the forward / backward check).  fp32 path (the parity mode) against the reference: flows
    <= 1e-4 px, occlusion masks, the four loss terms <= 2e-4, every gradient norm <= 1e-2, directions >= 0.9999."""
    g = load_golden('train_128x416_hs1')
    out, terms, names, params = _realistic_step('fp32')
    print('regime (reference): mean |flow| %.2f px, masked %.1f %%, samples outside the crop %.1f %%, outside the frame %.1f %%'
          % (float(g['regime'][0]), 100 * float(g['regime'][1]), 100 * float(g['regime'][2]), 100 * float(g['regime'][3])))
    assert oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out']) <= 1e-4
    assert oracle.epe(out['flow_b_out'].detach().cpu(), g['flow_b_out']) <= 1e-4
    for k_ in ('occ_fw', 'occ_bw'):
        assert (out[k_].detach().cpu() != g[k_].float()).float().mean() <= 2e-3, k_
    for k, v in terms.items():
        want = float(g[k])
        print(k, float(v), want)
        assert abs(float(v) - want) <= 2e-4 * max(1.0, abs(want)), k
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    cos, worst = grad_direction_check({n: params[n].grad for n in names}, g)
    print('realistic motion, fp32: max rel grad-norm error %.3g (param %s), min cosine %.7f, worst bias-gradient error %.3g'
          % (rel.max(), names[int(rel.argmax())], cos.min(), worst))
    # (measured over boxes: norms 2.6e-3 .. 5.3e-3, worst bias gradient 4e-3 .. 8e-3, cosines >= 0.99996.  The fp32 training path runs
    # its convolutions through PyTorch-ROCm, whose gradient kernels are not reproducible run to run, and at this motion 94 % of the
    # pixels sit behind hard masks: the reference's OWN gradients move to cosine 0.983 when its frames are rounded to bf16.)
    assert (got > 0).all() and rel.max() <= 1e-2 and cos.min() >= 0.9999 and worst <= 1.5e-2


# measured on MI355X (printed by the test): loss terms within 0.52 %, flow 0.83 px = 7.3 % of the motion (128x416 with random full-scale
# heads is a chaotic map: the bf16 INFERENCE path is 2.9 % away at 256x256, 1.1 % at 384x1280), gradient norms within 14.8 % (median 4.1 %),
# gradient cosines min 0.910 / median 0.985 against 0.983 / 0.998 for the reference's own step with bf16-rounded frames — i.e. at
# realistic motion the bf16 training mode IS further from the reference than an input-rounding-sized perturbation; the small-motion
# vector (0.2 % / 4 %, on the floor) did not show that.  Bounds = 1.5 x the measurement.
BF16_HS1 = {'loss': 8e-3, 'epe': 1.25, 'gnorm': 0.22, 'cos_median_slack': 0.02, 'cos_min_slack': 0.11}


def test_bf16_training_mode_at_realistic_motion_stays_in_the_envelope():
    """The bf16 training mode on the same vector: the envelope of test_bf16_training_mode_tracks_the_fp32_reference RESTATED where
    the masks are active and the warps leave the crop (BF16_HS1 above: the measured distances and what they say)."""
    g = load_golden('train_128x416_hs1')
    out, terms, names, params = _realistic_step('bf16')
    lrel = {k: abs(float(v) - float(g[k])) / max(1.0, abs(float(g[k]))) for k, v in terms.items()}
    print('loss terms, relative error:', {k: round(v, 5) for k, v in lrel.items()})
    e = oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out'])
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    cos, worst = grad_direction_check({n: params[n].grad for n in names}, g)
    ref, r16 = g['grad_proj'].numpy(), g['grad_proj_bf16_frames'].numpy()
    floor = (ref * r16).sum(1) / (np.linalg.norm(ref, axis=1) * np.linalg.norm(r16, axis=1))
    print('realistic motion, bf16: loss terms rel. error max %.4f, flow EPE %.4f px, max rel grad-norm error %.3g (param %s) median %.3g, cosine min %.5f median %.5f | reference with bf16 frames: min %.5f median %.5f'
          % (max(lrel.values()), e, rel.max(), names[int(rel.argmax())], np.median(rel), cos.min(), np.median(cos), floor.min(), np.median(floor)))
    assert max(lrel.values()) <= BF16_HS1['loss'] and e <= BF16_HS1['epe'] and rel.max() <= BF16_HS1['gnorm']
    assert np.median(cos) >= np.median(floor) - BF16_HS1['cos_median_slack'] and cos.min() >= floor.min() - BF16_HS1['cos_min_slack']


def test_fp16_training_mode_at_realistic_motion_meets_the_tight_bars():
    """VERDICT r5 next 3 (gradient cosine min >= 0.97, norms <= 6 % at realistic motion).  Round 6 localised the bf16 mode's distance on this
    vector (profiles/r06_train_precision_localise.txt, r06_train_forward_trace.txt): bf16 rounding injected at EVERY tensor class of the fp32
    step moves the flow by 0.07 px and the worst cosine to 0.970 — the 0.83 px / 0.910 of the real bf16 forward is ONE validity-mask flip of
    the feature warp at the 4x13 level (1 pixel of 208; then 6, 12, 45 at the finer levels), a discrete event of the reference's hard masks
     that any perturbation of that size can trigger, not a precision defect of one tensor class.  The fp16 matrix-core mode
    (`train_conv_dtype='fp16'`, 11 mantissa bits; measured: flow 0.012 px, loss terms 1.2e-4, gradient norms <= 3.2 % (median 0.5 %),
    cosines min 0.997 / median 0.9998) stays on the reference's side of every mask on this vector and meets the bars."""
    g = load_golden('train_128x416_hs1')
    out, terms, names, params = _realistic_step('fp16')
    lrel = {k: abs(float(v) - float(g[k])) / max(1.0, abs(float(g[k]))) for k, v in terms.items()}
    e = oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out'])
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    cos, worst = grad_direction_check({n: params[n].grad for n in names}, g)
    print('realistic motion, fp16: loss terms rel. error max %.5f, flow EPE %.4f px, max rel grad-norm error %.3g median %.3g, cosine min %.5f median %.5f'
          % (max(lrel.values()), e, rel.max(), np.median(rel), cos.min(), np.median(cos)))
    assert all(torch.isfinite(params[n].grad).all() for n in names)
    assert max(lrel.values()) <= 1e-3 and e <= 0.05
    assert rel.max() <= 0.06 and cos.min() >= 0.97 and np.median(cos) >= 0.998
