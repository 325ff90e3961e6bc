"""runtime.PipelinedInference at the SMALL shape of tests/test_hip_net.py::test_pipelined_inference_equals_one_step_at_a_time
(2 x 128 x 256, three streams, three different batches): how often does a slot differ from the same batch run alone?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _weights
import test_hip_net as T
from upflow_pytorch_amd.runtime import GraphedInference, PipelinedInference
dev = torch.device('cuda', 0)
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "2x128x256,2x64x128,4x192x640").split(",")]
for dt in [getattr(torch, n) for n in os.environ.get('DTYPES', 'bfloat16,float16').split(',')]:
    net = T.build('robust', dt)
    for (B, H, W) in shapes:
        batches = [tuple(t.cuda() for t in _weights.make_smooth_images(40 + i, B, H, W)) for i in range(3)]
        single = GraphedInference(net, B, H, W, device=dev)
        ref = []
        for a, b in batches:
            ref.append({k: v.clone() for k, v in single(a, b).items()})
        # the single runner itself, replayed again: deterministic on an idle GPU?
        again = sum(int(not torch.equal(single(a, b)[k], r[k])) for (a, b), r in zip(batches, ref) for k in r if torch.is_tensor(r[k]))
        pipe = PipelinedInference(net, B, H, W, streams=3, device=dev)
        for s, (a, b) in enumerate(batches):
            pipe.load(s, a, b)
        bad, first = 0, None
        rounds = 60
        for it in range(rounds):
            for s in range(3):
                pipe.replay(s)
            pipe.synchronize()
            for s in range(3):
                o = pipe.result(s)
                for k in ('flow_f_out', 'flow_b_out', 'occ_fw', 'occ_bw'):
                    if not torch.equal(o[k], ref[s][k]):
                        bad += 1
                        if first is None:
                            d = (o[k].float() - ref[s][k].float()).abs()
                            first = (it, s, k, float(d.max()), int((d > 0).sum()))
        print('%s [%d,%d,%d]: single replayed again differs %d; 3 streams: mismatching outputs %d of %d  first %s' % (
            str(dt).split('.')[-1], B, H, W, again, bad, rounds * 12, first), flush=True)
        del pipe, single
        torch.cuda.empty_cache()
