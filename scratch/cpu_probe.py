import os, sys, time, torch
sys.path.insert(0,"vidsgg-big_amd"); sys.path.insert(0,".")
from bigc_hip import configs
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal
from oracle import bigc_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
cfg=dict(configs.CONFIG_V); sd=seeded_state_dict(bigc_state_shapes(cfg,"vidvrd"),7,8.0)
p=[synthetic_proposal(1234+v,n_trajs=64,feat_width=2880) for v in range(2)]
for nt in (4,8,16,32,64):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.forward_test(sd,cfg,p[:1],10)
        t=time.perf_counter(); O.forward_test(sd,cfg,p,10); dt=(time.perf_counter()-t)/2
    print(nt,"threads: %.3f s/video"%dt, flush=True)
