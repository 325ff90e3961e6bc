"""PipelinedInference debug (round 4): three runners with the model's debug taps (net._taps) — the FIRST intermediate buffer that differs between concurrent runners (led to the cost volume)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
for rnd, (B, H, W, dn) in enumerate([(4, 384, 1280, 'bf16'), (8, 448, 1024, 'fp16'), (4, 384, 1280, 'bf16')]):
    dt = bench.DT[dn]
    net = bench.build_net(dt, dev)
    a, b = [t.to(dev) for t in synthetic.make_smooth_images(50, B, H, W)]
    single = GraphedInference(net, B, H, W, device=dev)
    ref = {k: v.clone() for k, v in single(a, b).items()}
    del single
    with torch.no_grad():
        eager = net({'im1': a, 'im2': b, 'if_loss': False})
    print('round %d %s: single == eager %s' % (rnd, dn, torch.equal(ref['flow_f_out'], eager['flow_f_out'])))
    runners, taps = [], []
    for i in range(3):
        net._taps = []
        r = GraphedInference(net, B, H, W, device=dev, warmup=3 if i == 0 else 1)
        n = len(net._taps) // (4 if i == 0 else 2)
        taps.append(net._taps[-n:])
        runners.append(r)
        r.load(a, b)
    net._taps = None
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    for it in range(3):
        for r, s in zip(runners, streams):
            with torch.cuda.stream(s):
                r.replay()
        torch.cuda.synchronize()
    for i, r in enumerate(runners):
        ok = torch.equal(r.out['flow_f_out'], ref['flow_f_out'])
        msg = ''
        if not ok:
            for (n0, t0), (n1, t1) in zip(taps[i], taps[(i + 1) % 3]):
                if t0.shape == t1.shape and not torch.equal(t0, t1):
                    d = (t0.float() - t1.float()).abs()
                    msg = 'first tap differing from runner %d: %s shape %s max %.3g #diff %d' % ((i + 1) % 3, n0, tuple(t0.shape), float(d.max()), int((d > 0).sum()))
                    break
        print('   runner %d == reference: %s  %s' % (i, ok, msg), flush=True)
    del runners, taps, net
    torch.cuda.empty_cache()
