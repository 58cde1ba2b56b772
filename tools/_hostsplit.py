import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import torch
import train_parent as tp
from osvos_pytorch_amd import train_common as tc
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
orig_mb = tc.TrainLoop.micro_batch
def mb(self, *a, **k):
    t = time.perf_counter(); r = orig_mb(self, *a, **k)
    if os.environ.get('HS_SYNC') == '1': torch.cuda.synchronize()
    acc['micro_batch'] += time.perf_counter() - t; cnt['micro_batch'] += 1; return r
tc.TrainLoop.micro_batch = mb
orig_fw = None
orig_es = tp.epoch_samples
def es(*a, **k):
    it = orig_es(*a, **k)
    while True:
        t = time.perf_counter()
        try: s = next(it)
        except StopIteration: return
        acc['next_sample'] += time.perf_counter() - t; cnt['next_sample'] += 1
        yield s
tp.epoch_samples = es
orig_bw = torch.autograd.backward
def bw(*a, **k):
    t = time.perf_counter(); r = orig_bw(*a, **k); acc['backward'] += time.perf_counter() - t; return r
torch.autograd.backward = bw
t0 = time.perf_counter()
tp.main(sys.argv[1:])
torch.cuda.synchronize()
print("total %.3f s" % (time.perf_counter() - t0))
for k in acc: print("%-12s %.3f ms per call over %d" % (k, acc[k] / max(1, cnt.get(k, cnt['micro_batch'])) * 1e3, cnt.get(k, cnt['micro_batch'])))
