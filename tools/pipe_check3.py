"""PipelinedInference debug (round 4): eager vs one graph vs three slots, sequential and concurrent replays, per input — separates input-loading faults from concurrency faults."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
B, H, W = 4, 384, 1280
net = bench.build_net(torch.bfloat16, dev)
ims = [tuple(t.to(dev) for t in synthetic.make_smooth_images(50 + i, B, H, W)) for i in range(3)]
eager = []
with torch.no_grad():
    for a, b in ims:
        eager.append({k: v.clone() for k, v in net({'im1': a, 'im2': b, 'if_loss': False}).items()})
single = GraphedInference(net, B, H, W, device=dev)
gr = [{k: v.clone() for k, v in single(a, b).items()} for a, b in ims]
for i in range(3):
    print('input %d: single graph == eager: %s' % (i, all(torch.equal(gr[i][k], eager[i][k]) for k in ('flow_f_out', 'flow_b_out'))))
# replay the single graph again on each input, twice
for rep in range(2):
    for i, (a, b) in enumerate(ims):
        o = single(a, b)
        torch.cuda.synchronize()
        print('  rep %d input %d: single graph again == eager: %s  max diff %.3g' % (rep, i, torch.equal(o['flow_f_out'], eager[i]['flow_f_out']), float((o['flow_f_out'] - eager[i]['flow_f_out']).abs().max())))
pipe = PipelinedInference(net, B, H, W, streams=3, device=dev)
for s, (a, b) in enumerate(ims):
    pipe.load(s, a, b)
for mode in ('sequential', 'concurrent'):
    for it in range(3):
        for s in range(3):
            pipe.replay(s)
            if mode == 'sequential':
                torch.cuda.synchronize()
        pipe.synchronize()
        print(mode, it, [(bool(torch.equal(pipe.result(s)['flow_f_out'], eager[s]['flow_f_out'])), round(float((pipe.result(s)['flow_f_out'] - eager[s]['flow_f_out']).abs().max()), 4)) for s in range(3)])
