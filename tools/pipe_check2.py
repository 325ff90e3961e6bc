"""PipelinedInference at the other workloads: concurrent steps vs a step run alone, bit for bit."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
for name, (B, H, W, dn) in list(bench.WORKLOADS.items()) + [('config2_fp32', (2, 384, 1280, 'fp32')), ('small', (2, 128, 256, 'bf16'))]:
    dt = bench.DT[dn]
    net = bench.build_net(dt, dev)
    ims = [synthetic.make_smooth_images(50 + i, B, H, W) for i in range(3)]
    single = GraphedInference(net, B, H, W, device=dev)
    refs = []
    for a, b in ims:
        refs.append({k: v.clone() for k, v in single(a.to(dev), b.to(dev)).items()})
    del single
    pipe = PipelinedInference(net, B, H, W, streams=3, device=dev)
    for s, (a, b) in enumerate(ims):
        pipe.load(s, a.to(dev), b.to(dev))
    bad = tot = 0
    detail = {}
    with torch.no_grad():
        eager = [net({'im1': a.to(dev), 'im2': b.to(dev), 'if_loss': False}) for a, b in ims]
    ref_vs_eager = [all(torch.equal(refs[s][k], eager[s][k]) for k in ('flow_f_out', 'flow_b_out')) for s in range(3)]
    for it in range(12):
        for s in range(3):
            pipe.replay(s)
        pipe.synchronize()
        for s in range(3):
            o = pipe.result(s)
            for k in ('flow_f_out', 'flow_b_out', 'occ_fw', 'occ_bw'):
                tot += 1
                ne = not torch.equal(o[k], refs[s][k])
                bad += int(ne)
                if ne:
                    detail[(s, k)] = (detail.get((s, k), (0, 0))[0] + 1, max(detail.get((s, k), (0, 0))[1], float((o[k].float() - refs[s][k].float()).abs().max())), bool(torch.equal(o[k], eager[s][k])))
    print('%-14s %dx%d B=%d %s: mismatching outputs %d of %d   single-graph reference == eager forward: %s   %s' % (name, H, W, B, dn, bad, tot, ref_vs_eager, detail), flush=True)
    del pipe, net
    torch.cuda.empty_cache()
