"""PipelinedInference debug (round 4): which model switches (_no_c8, _no_c8_est, _no_c8_sgu, fused normalisation off, narrow kernel off) make concurrent replays reproducible again."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
for dt in (torch.bfloat16, torch.float16):
    net = bench.build_net(dt, dev)
    B, H, W = 4, 384, 1280
    a, b = synthetic.make_smooth_images(2, 2, H, W)
    idx = [1, 0, 0, 1]
    a, b = a[idx].contiguous().to(dev), b[idx].contiguous().to(dev)
    single = GraphedInference(net, B, H, W, device=dev)
    ref = {k: v.clone() for k, v in single(a, b).items()}
    print(dt, 'single: items equal', torch.equal(ref['flow_f_out'][0], ref['flow_f_out'][3]), torch.equal(ref['flow_f_out'][1], ref['flow_f_out'][2]))
    with torch.no_grad():
        e = net({'im1': a, 'im2': b, 'if_loss': False})
    print(dt, 'eager == graph', torch.equal(e['flow_f_out'], ref['flow_f_out']))
    pipe = PipelinedInference(net, B, H, W, streams=2, device=dev)
    for s in range(2):
        pipe.load(s, a, b)
    for mode in ('sequential', 'concurrent'):
        bad = 0
        for it in range(10):
            for s in range(2):
                pipe.replay(s)
                if mode == 'sequential':
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            for s in range(2):
                o = pipe.result(s)
                for k in ('flow_f_out', 'flow_b_out'):
                    if not torch.equal(o[k], ref[k]):
                        bad += 1
                        if bad <= 4:
                            d = (o[k] - ref[k]).abs()
                            print('   ', mode, 'iter', it, 'slot', s, k, 'max diff %.3g' % float(d.max()), 'items differing', [int(i) for i in range(B) if float(d[i].max()) > 0])
        print(dt, mode, 'mismatching outputs:', bad, 'of 40')
    del pipe, single
