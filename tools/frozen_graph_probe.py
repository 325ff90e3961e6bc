#!/usr/bin/env python3
"""Debug harness: the frozen-decoder variant of test_captured_graph_survives_cache_clears_allocator_churn_and_an_eager_step under
different optimizer set-ups (round 4, fused Adam + version bumps)."""
import os
import sys
import warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.train import synthetic_train_batch
from upflow_pytorch_amd.utils import loss as loss_mod
import test_hip_train as T

batch = synthetic_train_batch(4, device='cuda')
small = {k: v[:1].contiguous() for k, v in batch.items()}


SKIP = set(os.environ.get('SKIP', '').split(','))


def churn(hold):
    if 'caches' not in SKIP:
        ops.train_caches_clear()
    if 'valid' not in SKIP:
        loss_mod._VALID.clear()
    if 'zeros' not in SKIP:
        loss_mod._ZEROS.clear()
    torch.cuda.synchronize()
    if 'empty' not in SKIP:
        torch.cuda.empty_cache()
    if 'fill' not in SKIP:
        for nbytes in [int(v) for v in os.environ.get('FILL', '512,4096,16384,65536,1048576,4194304').split(',')]:
            for _ in range(24):
                hold.append(torch.full((nbytes // 4,), float('nan'), dtype=torch.float32, device='cuda'))
    torch.cuda.synchronize()


def run(variant, do_churn):
    tr = T._config3_trainer('bf16', True)
    tr.raw_net.froze_PWC()
    live = [p for p in tr.net.parameters() if p.requires_grad]
    if variant == 'A':                                   # foreach optimizer, no version bumps
        tr.fused_adam = False
        tr.optimizer = torch.optim.Adam(live, lr=tr.optimizer.param_groups[0]['lr'], amsgrad=True, weight_decay=1e-4, capturable=True)
    elif variant == 'B':                                 # the test as written: foreach optimizer, bumps on ALL parameters
        tr.optimizer = torch.optim.Adam(live, lr=tr.optimizer.param_groups[0]['lr'], amsgrad=True, weight_decay=1e-4, capturable=True)
    elif variant == 'C':                                 # fused optimizer on the live parameters, bumps on the live parameters
        tr.optimizer = torch.optim.Adam(live, lr=tr.optimizer.param_groups[0]['lr'], amsgrad=True, weight_decay=1e-4, capturable=True, fused=True)
        tr._opt_params = live
    if variant in ('B2', 'B3'):                          # foreach optimizer; every parameter's version moved after every step body
        tr.fused_adam = False
        tr.optimizer = torch.optim.Adam(live, lr=tr.optimizer.param_groups[0]['lr'], amsgrad=True, weight_decay=1e-4, capturable=True)
        body = tr._step_body

        pinned = []

        def bumped(b):
            r = body(b)
            if variant == 'B3' and not torch.cuda.is_current_stream_capturing():
                pinned.extend(ops._train_cache_tensors())     # experiment: nothing cached before the capture is ever freed
            torch.autograd.graph.increment_version(list(tr.net.parameters()))
            return r
        tr._step_body = bumped
        tr._pinned = pinned
    hold, stats = [], []
    for _ in range(tr.graph_warmup + 1):
        stats.append(tr.step(batch))
    assert tr._graph is not None
    if do_churn:
        churn(hold)
    stats += [tr.step(batch) for _ in range(2)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        stats.append(tr.step(small))
    if do_churn:
        churn(hold)
    stats += [tr.step(batch) for _ in range(3)]
    global LAST_TR, HOLD
    LAST_TR = tr
    HOLD = hold
    return stats


for variant in ([] if (os.environ.get('INSPECT') or os.environ.get('POOLS') or os.environ.get('HISTORY')) else (sys.argv[1:] or ['A', 'B', 'C'])):
    a = run(variant, False)
    torch.cuda.empty_cache()
    b = run(variant, True)
    torch.cuda.empty_cache()
    print(variant, 'identical' if a == b else 'DIFFER at %s' % [i for i, (x, y) in enumerate(zip(a, b)) if x != y])
    if a != b:
        i = [i for i, (x, y) in enumerate(zip(a, b)) if x != y][0]
        print('   ', a[i], '\n   ', b[i])


def inspect():
    """Record the tensors the captured census loss reads; after a bad replay say which of them holds what."""
    rec = []
    cd, rl = ops.census_distance, ops.robust_loss_sums

    def cd_w(g1, g2, md=3):
        r = cd(g1, g2, md)
        if torch.cuda.is_current_stream_capturing():
            rec.append(('census_distance', dict(gray1=g1, gray2=g2, dist=r)))
        return r

    def rl_w(x, y, m, **kw):
        r = rl(x, y, m, **kw)
        if torch.cuda.is_current_stream_capturing():
            rec.append(('robust_loss_sums', dict(x=x, y=y, mask=m, out=r if torch.is_tensor(r) else list(r))))
        return r
    ops.census_distance, ops.robust_loss_sums = cd_w, rl_w
    gw = loss_mod._grey_weights

    def gw_w(like):
        w = gw(like)
        if torch.cuda.is_current_stream_capturing():
            rec.append(('grey_weights', dict(w=w)))
        return w
    loss_mod._grey_weights = gw_w
    cv = loss_mod._census_valid

    def cv_w(mask, md):
        v = cv(mask, md)
        if torch.cuda.is_current_stream_capturing():
            rec.append(('census_valid', dict(mask=mask, valid=v)))
        return v
    loss_mod._census_valid = cv_w
    try:
        stats = run(os.environ.get('INSPECT_VARIANT', 'B2'), os.environ.get('INSPECT_CHURN', '1') == '1')
    finally:
        ops.census_distance, ops.robust_loss_sums, loss_mod._grey_weights, loss_mod._census_valid = cd, rl, gw, cv
    print('last stats', stats[-1])

    def desc(t):
        if t is None:
            return 'None'
        if not torch.is_tensor(t):
            return str([desc(u) for u in t])
        f = t.detach().float()
        return '%s %s ptr %x nan %d min %.4g max %.4g mean %.4g' % (tuple(t.shape), t.dtype, t.data_ptr(), int(torch.isnan(f).sum()), float(torch.nan_to_num(f).min()), float(torch.nan_to_num(f).max()), float(torch.nan_to_num(f).mean()))
    global LAST_TR
    dists = [d['dist'] for n, d in rec if n == 'census_distance']
    vms = [(d['mask'], d['valid']) for n, d in rec if n == 'census_valid']
    for dd, (m, v) in zip(dists, vms):
        x = dd.detach().float()
        print('eager re-evaluation from the recorded tensors: mean((|d|+.01)^0.4) = %.5f   sum(d*m)/(2 sum m) = %.5f   charbonnier mean = %.5f' % (
            float((x.abs() + 0.01).pow(0.4).mean()), float(((x.abs() + 0.01).pow(0.4) * m * v).sum() / ((m * v).sum() * 2 + 1e-6)), float((x ** 2 + 1e-8).pow(0.4).mean())))
    ka = set(t.data_ptr() for t in LAST_TR._graph_keepalive)
    for name, d in rec:
        if name == 'census_valid':
            print('   valid pinned by the trainer:', d['valid'].data_ptr() in ka, ' refcount-holders: _VALID has it:', any(v is d['valid'] for v in loss_mod._VALID.values()))
        print(name)
        for k, v in d.items():
            print('    %-6s %s' % (k, desc(v)))


if os.environ.get('INSPECT'):
    inspect()


def pools():
    """Where do eager allocations made AFTER the capture land?  (segment_pool_id (0, 0) = the default pool)"""
    global HOLD
    HOLD = []
    run('B2', True)
    snap = torch.cuda.memory_snapshot()
    segs = [(s['address'], s['address'] + s['total_size'], tuple(s.get('segment_pool_id', (0, 0)))) for s in snap]
    from collections import Counter
    c = Counter()
    for t in HOLD:
        p = t.data_ptr()
        pool = [sp for (a, b, sp) in segs if a <= p < b]
        c[pool[0] if pool else None] += 1
    print('pools of the %d NaN-filled tensors:' % len(HOLD), dict(c))
    print('segments per pool:', dict(Counter(sp for (_, _, sp) in segs)))


if os.environ.get('POOLS'):
    pools()


def history():
    """Default-pool blocks allocated BEFORE the end of the capture and freed AFTER it (during the eager partial step / the churn):
    candidates for what the graph still reads.  Prints where each was allocated."""
    torch.cuda.memory._record_memory_history(max_entries=400000, context='alloc', stacks='python')
    marks = {}

    def mark(name, nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
        marks[name] = nbytes
        del t

    tr = T._config3_trainer('bf16', True)
    tr.raw_net.froze_PWC()
    live = [p for p in tr.net.parameters() if p.requires_grad]
    tr.optimizer = torch.optim.Adam(live, lr=tr.optimizer.param_groups[0]['lr'], amsgrad=True, weight_decay=1e-4, capturable=True)
    body = tr._step_body

    def bumped(b):
        r = body(b)
        torch.autograd.graph.increment_version(list(tr.net.parameters()))
        return r
    tr._step_body = bumped
    tcm = ops.train_caches_mark

    def tcm_w():
        mark('capture_start', 6666)
        return tcm()
    ops.train_caches_mark = tcm_w
    for i in range(tr.graph_warmup):
        if i == tr.graph_warmup - 1:
            mark('before_last_warmup', 7777)
        tr.step(batch)
    ops.train_caches_mark = tcm
    assert tr._graph is not None
    mark('after_capture', 8888)
    tr.step(batch); tr.step(batch)
    mark('before_eager', 9999)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tr.step(small)
    hold = []
    churn(hold)
    mark('after_eager', 11111)
    snap = torch.cuda.memory._snapshot()
    default_segs = [(sg['address'], sg['address'] + sg['total_size']) for sg in snap['segments'] if tuple(sg.get('segment_pool_id', (0, 0))) == (0, 0)]
    priv_segs = [(sg['address'], sg['address'] + sg['total_size']) for sg in snap['segments'] if tuple(sg.get('segment_pool_id', (0, 0))) != (0, 0)]
    torch.cuda.memory._record_memory_history(enabled=None)
    ev = snap['device_traces'][0]
    idx = {}
    for i, e in enumerate(ev):
        if e['action'] == 'alloc':
            for name, nb in marks.items():
                if e['size'] == nb and name not in idx:
                    idx[name] = i
    print('marks at', idx, 'of', len(ev), 'events')
    live_alloc = {}
    out = []
    for i, e in enumerate(ev):
        if e['action'] == 'alloc':
            live_alloc[e['addr']] = (i, e)
        elif e['action'] in ('free_requested', 'free'):
            a = live_alloc.pop(e['addr'], None)
            if a is not None and a[0] < idx['after_capture'] and idx['before_eager'] < i < idx['after_eager']:
                if not any(lo <= e['addr'] < hi for lo, hi in priv_segs):       # (not the graph's private pool)
                    out.append((a[0], i, a[1]))
    print(len(out), 'DEFAULT-pool blocks allocated before the capture ended and freed during the eager partial step / the churn')
    out2 = []
    live_alloc = {}
    for i, e in enumerate(ev):
        if e['action'] == 'alloc':
            live_alloc[e['addr']] = (i, e)
        elif e['action'] in ('free_requested', 'free'):
            a = live_alloc.pop(e['addr'], None)
            if a is not None and a[0] < idx['capture_start'] and idx['capture_start'] < i < idx['after_capture'] and not any(lo <= e['addr'] < hi for lo, hi in priv_segs):
                out2.append((a[0], i, a[1]))
    in_cap = [(i, e) for i, e in enumerate(ev) if e['action'] == 'alloc' and idx['capture_start'] < i < idx['after_capture']
              and any(lo <= e['addr'] < hi for lo, hi in default_segs) and e['size'] not in (6666, 8888)]
    print(len(in_cap), 'allocations made DURING the capture that landed in the DEFAULT pool:')
    for i, e in in_cap[:40]:
        fr = [f for f in e.get('frames', []) if 'upflow_pytorch_amd' in f.get('filename', '') or 'tools/' in f.get('filename', '')]
        print('   event %6d size %9d  %s' % (i, e['size'], ' <- '.join('%s:%d %s' % (f['filename'].split('/')[-1], f['line'], f['name']) for f in fr[:6])))
    print(len(out2), 'DEFAULT-pool blocks allocated before the capture STARTED and freed DURING it:')
    seen = set()
    for (ai, fi, e) in out2 + out:
        fr = [f for f in e.get('frames', []) if 'upflow_pytorch_amd' in f.get('filename', '') or 'tools/' in f.get('filename', '') or 'tests/' in f.get('filename', '')]
        key = (e['size'], tuple((f['filename'].split('/')[-1], f['line']) for f in fr[:4]))
        if key in seen:
            continue
        seen.add(key)
        when = 'warm-up' if ai < idx['before_last_warmup'] else ('last warm-up step or capture')
        print('   size %9d  allocated at event %6d (%s), freed at %6d:  %s' % (e['size'], ai, when, fi, ' <- '.join('%s:%d %s' % (f['filename'].split('/')[-1], f['line'], f['name']) for f in fr[:5])))


if os.environ.get('HISTORY'):
    history()
