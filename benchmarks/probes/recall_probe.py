"""Tuning of the Recall@50 gate's difficulty (bench.py recall leg): R@50 of the oracle / HIP / f16 / bf16 for a few settings."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vidsgg-big_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bigc_hip import _lib, configs, evaluate as E, recall_set as R
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict
from oracle import bigc_oracle as O
dev = torch.device("cuda:0"); lib = _lib.load()
cfg = dict(configs.CONFIG_V)
for (noise, motion, actors, marker, steps, n_train) in [(0.15, True, 5, 2.0, 300, 128), (0.2, True, 5, 2.0, 300, 128), (0.25, True, 5, 2.0, 200, 128), (0.15, True, 5, 2.0, 400, 192)]:
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 11, att_gain=1.0)
    torch.manual_seed(1234)
    trainee = bench.build_model(cfg, sd, dev, "vidvrd", is_train=True)
    kw = dict(n_trajs=64, n_actors=actors, marker=marker, label_noise=noise, motion=motion)
    data = R.relational_set(cfg, 20000, n_train, **kw)
    for p, _ in data: p.to(dev)
    t0 = time.time(); loss = R.train(trainee, data, steps, batch=16, lr=1e-4); torch.cuda.synchronize(); tt = time.time() - t0
    sd_t = {k: v.detach().cpu() for k, v in trainee.state_dict().items()}
    del trainee, data; torch.cuda.empty_cache()
    model = bench.build_model(cfg, sd_t, dev, "vidvrd", is_train=False)
    held = R.relational_set(cfg, 30000, 16, **kw)
    gts = R.ground_truth(held); props = [p for p, _ in held]
    dprops = [copy.deepcopy(p).to(dev) for p in props]
    rel = {"hip": R.relations(props, model(dprops, topk=10))}
    for name, mode in (("f16", 1), ("bf16", 2)):
        lib.bigc_linear_set_fast(mode)
        try: rel[name] = R.relations(props, model(dprops, topk=10))
        finally: lib.bigc_linear_set_fast(0)
    torch.set_num_threads(16)
    with torch.no_grad(): rel["oracle"] = R.relations(props, O.forward_test(sd_t, cfg, props, 10))
    res = {k: E.evaluate(gts, v) for k, v in rel.items()}
    print("noise %.2f motion %s actors %d marker %.1f steps %d n_train %d loss %.3f train %.1fs:" % (noise, motion, actors, marker, steps, n_train, loss, tt),
          {k: (round(v[1][50], 4), round(v[1][100], 4), round(v[0], 4)) for k, v in res.items()}, flush=True)
