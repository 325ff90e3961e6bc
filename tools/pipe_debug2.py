"""PipelinedInference debug (round 4): as pipe_debug.py, the switches one at a time with mismatch counts per output."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
B, H, W = 4, 384, 1280
a, b = synthetic.make_smooth_images(2, 2, H, W)
idx = [1, 0, 0, 1]
a, b = a[idx].contiguous().to(dev), b[idx].contiguous().to(dev)

def trial(name, dt, toggles=(), two_nets=False, hs=0.1):
    nets = [bench.build_net(dt, dev) for _ in range(2 if two_nets else 1)]
    for n in nets:
        for t in toggles:
            setattr(n, t, True)
    single = GraphedInference(nets[0], B, H, W, device=dev)
    ref = {k: v.clone() for k, v in single(a, b).items()}
    if two_nets:
        pipe = PipelinedInference(nets[0], B, H, W, streams=1, device=dev)
        p2 = PipelinedInference(nets[1], B, H, W, streams=1, device=dev)
        runners = [pipe.runners[0], p2.runners[0]]
        streams = [pipe.streams[0], p2.streams[0]]
    else:
        pipe = PipelinedInference(nets[0], B, H, W, streams=2, device=dev)
        runners, streams = pipe.runners, pipe.streams
    for r in runners:
        r.load(a, b)
    torch.cuda.synchronize()
    bad = 0
    for it in range(15):
        for r, s in zip(runners, streams):
            with torch.cuda.stream(s):
                r.replay()
        torch.cuda.synchronize()
        for r in runners:
            for k in ('flow_f_out', 'flow_b_out'):
                bad += int(not torch.equal(r.out[k], ref[k]))
    print('%-40s %s mismatching outputs: %d of 60' % (name, str(dt).split('.')[-1], bad), flush=True)

import upflow_pytorch_amd.model.pwc_modules as pm
trial('bf16 base', torch.bfloat16)
trial('bf16 no_c8_est', torch.bfloat16, ('_no_c8_est',))
trial('bf16 no_c8_est no_c8_sgu', torch.bfloat16, ('_no_c8_est', '_no_c8_sgu'))
trial('bf16 no_c8_est no_c8_ctx', torch.bfloat16, ('_no_c8_est', '_no_c8_ctx'))
trial('bf16 no_c8_est no_c8_sgu no_c8_ctx', torch.bfloat16, ('_no_c8_est', '_no_c8_sgu', '_no_c8_ctx'))
pm._NO_NARROW[0] = True
trial('bf16 no_narrow', torch.bfloat16)
trial('bf16 no_c8_est no_narrow', torch.bfloat16, ('_no_c8_est',))
pm._NO_NARROW[0] = False
trial('bf16 no_fused_norm (c8 sgu+ctx on)', torch.bfloat16, ('_no_fused_norm',))
trial('bf16 base again', torch.bfloat16)
