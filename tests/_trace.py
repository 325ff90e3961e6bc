"""Teacher-forced replay (protocol P2) of the recorded reference forward.

tests/golden/trace_64x128.{npz,json} hold inputs and outputs of EVERY hot-op call of one reference
forward (64x128 pair, literal `mask >= 1.0` semantics) in call order.  `replay(provider)` feeds each
call's recorded inputs to `provider` (the oracle on CPU, or the HIP ops on the GPU) and compares
with the recorded outputs, so no chaos can build up between calls.
"""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def load_trace():
    events = json.load(open(os.path.join(GOLDEN, 'trace_64x128.json')))
    z = np.load(os.path.join(GOLDEN, 'trace_64x128.npz'))
    return events, z


def replay(provider, to_dev=lambda t: t, to_cpu=lambda t: t):
    """-> dict op -> list of max-abs errors, plus 'final_epe' (EPE of the two final flows produced
    from recorded inputs) and 'mask_mismatch' (count of warp-mask bit differences)."""
    events, z = load_trace()
    errs = {}
    mask_mismatch = 0
    final = []

    def T(ev, k):
        return to_dev(torch.from_numpy(z['e%03d_%s' % (ev['idx'], k)]))

    def add(op, got, want):
        got = to_cpu(got).float()
        errs.setdefault(op, []).append(float((got - want).abs().max()))

    for ev in events:
        op = ev['op']
        W = lambda k: torch.from_numpy(z['e%03d_%s' % (ev['idx'], k)])  # noqa: E731
        if op == 'corr':
            add(op, provider.corr81(T(ev, 'f1'), T(ev, 'f2')), W('out'))
        elif op in ('warp_mask', 'warp'):
            y = provider.warp(T(ev, 'x'), T(ev, 'flow'), 'literal' if op == 'warp_mask' else None)
            y = to_cpu(y).float()
            want = W('y')
            if op == 'warp_mask':
                # a wrong mask bit shows up as a zeroed/un-zeroed pixel: count them separately
                mm = ((y == 0) != (want == 0)).any(dim=1)
                mask_mismatch += int(mm.sum())
            errs.setdefault(op, []).append(float((y - want).abs().max()))
        elif op in ('upsample_rate', 'upsample'):
            want = W('y')
            add(op, provider.flow_upsample(T(ev, 'x'), want.shape[2], want.shape[3], op == 'upsample_rate'), want)
        elif op == 'normalize':
            na, nb = provider.normalize_pair(T(ev, 'a'), T(ev, 'b'))
            add(op, na, W('na'))
            add(op, nb, W('nb'))
        elif op == 'sgu_blend':
            olf = T(ev, 'output_level_flow') if 'output_level_flow' in ev['keys'] else None
            _, up, inter_flow, inter_mask = provider.sgu_blend(T(ev, 'flow_init'), T(ev, 'x_out'), olf)
            add('sgu_blend', up, W('flow_up'))
            add('sgu_inter', inter_flow, W('inter_flow'))
            add('sgu_mask', inter_mask, W('inter_mask'))
            if olf is not None:
                final.append((to_cpu(up).float(), W('flow_up')))
        else:
            raise KeyError(op)
    epes = [float((a.double() - b.double()).pow(2).sum(1).sqrt().mean()) for a, b in final]
    return {'errs': errs, 'mask_mismatch': mask_mismatch, 'final_epe': max(epes), 'n_events': len(events)}
