"""Where does the host time of PackedBatch(proposals, device, "lazy") go? (cProfile, 256 block-backed videos on the GPU)"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from bigc_hip.packed import PackedBatch
dev = torch.device("cuda:0")
props = [bench.device_proposal(1234 + v, 64, 2880, 36, (8, 64), (0, 64), dev) for v in range(256)]
for _ in range(5):
    PackedBatch(props, dev, "lazy")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    pk = PackedBatch(props, dev, "lazy")
torch.cuda.synchronize()
print("PackedBatch lazy: %.3f ms" % ((time.perf_counter() - t) * 50))
t = time.perf_counter()
for _ in range(20):
    pk = PackedBatch(props, dev, "lazy"); pk.rows_in_place(); pk.boxes
torch.cuda.synchronize()
print("+ rows_in_place + boxes: %.3f ms" % ((time.perf_counter() - t) * 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    pk = PackedBatch(props, dev, "lazy"); pk.rows_in_place(); pk.boxes
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(18)
