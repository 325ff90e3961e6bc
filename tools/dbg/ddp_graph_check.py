"""1-rank RCCL DDP + hipGraph capture of the training step: does it run, and how fast?"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dist.init_process_group(backend='nccl', rank=0, world_size=1)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
net = conf(); net.load_state_dict(synthetic.make_state_dict(0, head_scale=0.1))
dev = torch.device('cuda', 0)
tr = Trainer(net, device=dev, graph=True)
batch = synthetic_train_batch(4, seed=0, device=dev)
for i in range(tr.graph_warmup + 2):
    st = tr.step(batch)
print('graph captured:', tr._graph is not None, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.step(batch, sync_stats=False)
torch.cuda.synchronize(); print('DDP(1 rank, RCCL) + hipGraph: %.2f ms/step' % ((time.perf_counter() - t0) * 100))
dist.destroy_process_group()
