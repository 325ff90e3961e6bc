"""runtime.PipelinedInference: are concurrent steps bit-identical to a step run alone?  (bf16 and fp16, config-2 shape, 2-4 streams)"""
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
for dt in (torch.bfloat16, torch.float16):
    net = bench.build_net(dt, dev)
    single = GraphedInference(net, B, H, W, device=dev)
    ref = {k: v.clone() for k, v in single(a, b).items()}
    for ns in (2, 3, 4):
        pipe = PipelinedInference(net, B, H, W, streams=ns, device=dev)
        for s in range(ns):
            pipe.load(s, a, b)
        bad = 0
        for it in range(25):
            for s in range(ns):
                pipe.replay(s)
            pipe.synchronize()
            for s in range(ns):
                o = pipe.result(s)
                bad += sum(int(not torch.equal(o[k], ref[k])) for k in ('flow_f_out', 'flow_b_out', 'occ_fw', 'occ_bw'))
        print('%s %d streams: mismatching outputs %d of %d' % (str(dt).split('.')[-1], ns, bad, 25 * ns * 4), flush=True)
        del pipe
