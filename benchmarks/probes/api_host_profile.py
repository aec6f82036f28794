"""Where does the host time of one model(proposal_list, topk) call go?  (cProfile over 20 calls, 256 block-backed videos on the GPU, exp2 dims)
    python benchmarks/probes/api_host_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd")); sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1]
import torch
import bench
from bigc_hip import configs
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal
dev = torch.device("cuda:0")
cfg = dict(configs.CONFIG_V)
sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
width = cfg["dim_feat"] + cfg["dim_i3d"]
props = [synthetic_proposal(1234 + v, n_trajs=64, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(dev) for v in range(256)]
model = bench.build_model(cfg, sd, dev)
for _ in range(4):
    model(props, topk=10)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    model(props, topk=10)
torch.cuda.synchronize()
print("model(props): %.3f ms per call" % ((time.perf_counter() - t) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    model(props, topk=10)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(28)
