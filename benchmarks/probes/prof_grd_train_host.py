import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "benchmarks")); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["bench.py", "--gpus", "1"]
import torch
import bench
from legs import train as T
from legs.common import _lib
lib = _lib.load()
device = torch.device("cuda", 0)
# reuse the leg's construction by monkeypatching timed_with_events to profile the step
def fake_timed(lib_, step, warm, steps, repeats=3):
    for _ in range(4): step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    for _ in range(8): step()
    pr.disable(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host-issue ms per step %.2f, to GPU-done %.2f" % (1e3 * (t1 - t0) / 8, 1e3 * (t2 - t0) / 8))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
    raise SystemExit(0)
T.timed_with_events = fake_timed
T.grd_train_leg(bench.ARGS, lib, device)
