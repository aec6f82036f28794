"""The N > 1 paths of the PRODUCT on the GPU, with two ranks sharing the box's one GPU and the gloo
backend (RCCL cannot put two ranks on one device; the collective itself -- one all-reduce of the flat gradient buffer -- is
backend-agnostic and runs on RCCL in tests/test_tools_gpu.py::test_bench_train_mode_runs_rccl_allreduce_on_one_rank):
  * tools/eval_vidvrd.py and tools/eval_vidor.py under `torch.distributed.run --nproc-per-node 2`: size-balanced shares of the videos,
    results gathered on rank 0 -> the same metrics and the same per-video results as the one-process run;
  * the HIP training step data-parallel: two processes x B/2 videos with the flat-gradient all-reduce == what the reference's
    DataParallel computes (per-replica loss means, `.mean()` over replicas: utils/DataParallel.py:65-92, tools/train_vidvrd.py:155)."""
import os
import pickle
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "vidsgg-big_amd", "tools")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_tool(args, nproc=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc > 1:
        env.update(BIGC_DIST_BACKEND="gloo", BIGC_DIST_ONE_GPU="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=TOOLS, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def metrics(log):
    return [l for l in log.splitlines() if l.startswith(("detection mean AP", "detection recall", "tagging precision"))]


def same_results(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert (a[k] is None) == (b[k] is None), k
        if a[k] is not None:
            assert all(torch.equal(x, y) for x, y in zip(a[k], b[k])), k


def test_eval_vidvrd_two_ranks_equal_one_process():
    cfg = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp2", "config_.py")
    outs = []
    for nproc in (1, 2):
        d = tempfile.mkdtemp()
        log = run_tool([os.path.join(TOOLS, "eval_vidvrd.py"), "--cfg_path", cfg, "--output_dir", d, "--save_tag", "t", "--synthetic", "7",
                        "--n_trajs", "12"], nproc)
        assert "on %d GPU(s)" % nproc in log
        with open(os.path.join(d, "VidVRDtest_infer_result_t.pkl"), "rb") as f:
            outs.append((metrics(log), pickle.load(f)))
    assert len(outs[0][0]) == 3 and outs[0][0] == outs[1][0]
    same_results(outs[0][1], outs[1][1])


def test_eval_vidor_both_stages_two_ranks_equal_one_process():
    cfg5 = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "exp5", "config_.py")
    cfgg = os.path.join(ROOT, "vidsgg-big_amd", "experiments", "grounding_weights", "config_.py")
    res = []
    for nproc in (1, 2):
        d = tempfile.mkdtemp()
        log1 = run_tool([os.path.join(TOOLS, "eval_vidor.py"), "--eval_cls_only", "--cfg_path", cfg5, "--output_dir", d, "--save_tag", "t",
                         "--synthetic", "5", "--n_trajs", "10"], nproc)
        pkl = os.path.join(d, "VidORval_infer_results_topk3_t.pkl")
        with open(pkl, "rb") as f:
            raw = pickle.load(f)
        log2 = run_tool([os.path.join(TOOLS, "eval_vidor.py"), "--cfg_path", cfgg, "--cls_stage_result_path", pkl, "--output_dir", d, "--save_tag", "t",
                         "--synthetic", "5", "--n_trajs", "10", "--save_json_results"], nproc)
        with open(os.path.join(d, "VidORval_predict_relations_aft_grd_t.json")) as f:
            grounded = f.read()
        res.append((metrics(log1), raw, metrics(log2), grounded))
    assert res[0][0] == res[1][0] and len(res[0][0]) == 3
    same_results(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and len(res[0][2]) == 3
    import json
    a, b = json.loads(res[0][3]), json.loads(res[1][3])
    assert a.keys() == b.keys() and all(a[k] == b[k] for k in a)


def _dp_step_equals_reference_semantics(name, batch, backend):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from bigc_hip.shard import shard_indices
    from test_hip_train import build_train_model
    out = os.path.join(tempfile.mkdtemp(), "dp.pt")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), name, out, str(batch), backend], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    got = torch.load(out)
    assert got["in_place"] is True                       # the buffer the HIP backward filled was reduced in place
    # the reference semantics in one process: every chunk on its own, gradients averaged over the chunks
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    props, gts = props[:batch], gts[:batch]
    chunks = [c for c in (shard_indices(batch, r, 2) for r in range(2)) if c]
    want = None
    for c in chunks:
        model = build_train_model(cfg, variant, sd)
        g = cases.make_train_case(name)[4][:batch]
        loss, _ = model([props[i] for i in c], [g[i] for i in c])
        loss.mean().backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().cpu() / len(chunks) for k, p in model.named_parameters() if p.grad is not None}
        want = grads if want is None else {k: want[k] + grads[k] for k in want}
    assert got["grads"].keys() == want.keys()
    gmax = max(float(v.abs().max()) for v in want.values())
    for k in want:
        err = float((got["grads"][k] - want[k]).abs().max())
        assert err <= 1e-5 * max(float(want[k].abs().max()), 1e-4 * gmax), (k, err)


@pytest.mark.parametrize("name,batch", [("bigc_train_v64", 2), ("bigc_train_v64", 1), ("bigo_train_o", 2)])
def test_data_parallel_hip_training_step_equals_the_reference_semantics(name, batch):
    """two ranks (two processes on the one GPU, gloo) x their chunk of the batch + the flat-gradient all-reduce, against ONE process
    that evaluates the chunks one after the other and averages: per-replica loss means, mean over replicas (tools/train_vidvrd.py:155).
    batch = 1: the second rank has no chunk, contributes zeros, and the mean is over the one replica that exists.
    ``bigc_train_v64``: exp2 dims (VidVRD), two videos of N = 64 / 40; ``bigo_train_o``: exp5 dims (model_0v7, BASELINE configs[4]: VidOR
    DP training)."""
    _dp_step_equals_reference_semantics(name, batch, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with more than one rank needs two GPUs (one per rank); this box shows %d" % torch.cuda.device_count())
@pytest.mark.parametrize("name", ["bigc_train_v64", "bigo_train_o"])
def test_data_parallel_hip_training_step_on_two_gpus_over_rccl(name):
    """the same equivalence with the all-reduce on RCCL (backend "nccl", one rank per device): runs by itself wherever two devices
    are visible -- fresh child processes, the group created before any GPU call"""
    _dp_step_equals_reference_semantics(name, 2, "nccl")
