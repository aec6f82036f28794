#!/usr/bin/env python3
"""Benchmark of the BIG classification hot path (BIG-C eval forward incl. construct_triplet).

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): tracklet-pairs/sec = videos/s x N(N-1), whole job.
Workload = BASELINE.json configs[1]: VidVRD-test classification-only, model_0v10 at the exp2
dimensions (PKU RoI 2048-d + I3D 832-d), synthetic videos of N=64 tracklets (SURVEY.md 8d recipe,
video v uses numpy RandomState(1234+v)), numpy-seeded weights.  One step = ONE CALL OF THE DROP-IN
SIGNATURE, `model(proposal_list, topk)` (models/model_0v10.py:359-388), on `--videos` (default 256)
device-resident TrajProposals per GPU: the ragged batch's tables and boxes (the feature rows are read in
place through a row-address table: no copy of them is made), the forward, construct_triplet, the
compaction to the reference's return types, the one D2H copy and the per-video result tuples are all
inside the timed region.  The same
forward on a batch that stays packed in HBM is reported beside it (`resident_packed_batch`).  N>1: one process per GPU (torch.distributed / RCCL only for the barrier
and the max-over-ranks timing -- videos are independent, no data-path collective), weak scaling.
Started without a torch.distributed.run environment, `--gpus N` (N>1) launches the N ranks itself
(a child `python -m torch.distributed.run`, started before this process touches the GPU).

Also printed in the same JSON line (rank 0; everything below runs OUTSIDE the timed region):
  roofline      dominant kernel = the GEMM family (f32 GEMMs on the f16 matrix cores with exactly split
                operands, 3 MFMAs per product): algorithmic flops of its launches / their summed
                duration measured with HIP events inside the library on the launch stream over a repeat
                of the timed steps, against the dense f16 MFMA peak / 3
  cpu_baseline  the CPU oracle (oracle/bigc_oracle.py, a torch-CPU restatement of the reference's
                op sequence: kind "port") timed on the host cores, N=1 only, on a bounded sample of the
                same videos
  oracle_parity the oracle outputs of that sample compared row for row with the HIP outputs of the
                same videos: triplet rows identical, predicate logits max relative difference; and a FLOAT64 evaluation
                of the oracle for every sampled video (`distance_to_float64`: max / mean over the videos of the HIP path's
                and the f32 oracle's distance to it, videos above 1e-4, the per-query-row figure)
  plain_accumulation   the same call with bigc_set_accum_policy(0) (one accumulator per element, largest tiles): what the default's
                blocked accumulation costs and what it buys (its own distance to float64)
  ranks         who took part: rank, host, device ordinal + UUID and own step time of every rank, gathered through the job's process group
  exact_f32_gemm_mode / fast_f16_gemm_mode   the same step with every GEMM on the exact f32 MFMA, and in the opt-in
                reduced-precision mode (one f16 MFMA per product) with its Recall@50 gate
  vidor / grounding / train   the other BASELINE configs on the same GPU (N=1 only): VidOR-sized
                classification (N=180 tracklets; 8 host-seeded videos also through the CPU oracle), the grounding scorer on its
                triplet queries (a roofline per kernel family covering the step, step GFLOP against SURVEY.md 8d, compulsory bytes,
                two of its videos through the CPU oracle), and the BIG-C training step (forward + Hungarian + backward + all-reduce
                + clip + Adam; GEMM, dW and attention-backward rooflines)
  batch_x2      the headline workload at twice the videos per step
  forward_api / resident_packed_batch / forward_api_pipelined   the timed call (= value), the same forward on a batch that stays packed
                in HBM (with its own GEMM roofline), and the call in two halves (model.forward_async / .result(), two calls in flight)
  api_host_ms, modes, timed_steps_ms   host side of one call measured alone, the BIGC_* switches in effect, every timed call's duration
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos", type=int, default=256, help="videos per GPU per step")
    ap.add_argument("--n-trajs", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--cpu-videos", type=int, default=256, help="videos of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="stop the CPU-baseline sample after this much oracle time")
    ap.add_argument("--f64-videos", type=int, default=256, help="of the CPU sample, the videos that are also evaluated in float64 (the truth the HIP and the "
                                                                 "f32-oracle logits are measured against; ~0.15 s per video, outside every timing)")
    ap.add_argument("--headline", choices=["api", "packed"], default="api", help="what `value` times: api = model(proposal_list, topk) on device-resident "
                    "TrajProposals (pack + forward + results: the drop-in call), packed = the forward on a resident PackedBatch")
    ap.add_argument("--no-prof", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--mode", choices=["eval", "train"], default="eval", help="eval = headline metric; train = training-step throughput")
    ap.add_argument("--train-videos", type=int, default=8, help="videos per GPU per training step (reference batch size 8)")
    ap.add_argument("--exact-steps", type=int, default=3, help="extra steps timed with the exact f32-MFMA GEMM kernel, reported beside (0 = skip)")
    ap.add_argument("--fast-steps", type=int, default=3, help="extra steps timed in the opt-in reduced-precision mode (one f16 MFMA per product), "
                                                             "reported beside with its Recall@50 gate (0 = skip)")
    ap.add_argument("--recall-steps", type=int, default=300, help="training iterations of the Recall@50 leg (0 = skip the leg)")
    ap.add_argument("--recall-videos", type=int, default=16, help="held-out videos the Recall@50 leg scores (HIP, HIP reduced precision, CPU oracle)")
    ap.add_argument("--legs", default="vidor,grounding,train,batch512", help="extra workloads reported in the same line at N=1 ('' = none)")
    ap.add_argument("--vidor-videos", type=int, default=16, help="videos of the VidOR leg (SURVEY.md 8d: 16 x N = 180, L in [15, 1000))")
    ap.add_argument("--vidor-oracle-videos", type=int, default=8, help="of those, host-seeded ones that are also run through the CPU oracle (outside the timing)")
    ap.add_argument("--grd-oracle-videos", type=int, default=2, help="videos of the grounding leg that are also run through the CPU oracle (outside the timing)")
    ap.add_argument("--grd-one-video-per-call", action="store_true", help="grounding leg: the reference's one-video forward instead of the batched one")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `torch.distributed.run` (never an
    exec, and before this process has initialised the GPU), relay its output, return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


ARGS = parse() if __name__ == "__main__" else None
if ARGS is not None and (ARGS.gpus > 1 or os.environ.get("BIGC_BENCH_SELF_LAUNCH") == "1") and "RANK" not in os.environ:
    sys.exit(self_launch(ARGS))

import ctypes  # noqa: E402
import tempfile  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from bigc_hip import _lib, configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import TrajProposal, bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0   # same table, "Peak BF16/FP16 MFMA" dense
HBM_PEAK_GBS = 8000.0           # same table, HBM3E peak
EXACT_F32 = os.environ.get("BIGC_LINEAR_EXACT", "0") not in ("", "0")     # run the whole benchmark on the exact f32-MFMA GEMM kernel


def throttled_periods():
    """scheduling periods in which the cgroup's CPU quota froze this container so far (None without cgroup v2 accounting)"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except OSError:
        pass
    return None


def usable_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes report 256 CPUs but run under a 16-core quota; 256 threads there is ~900x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def build_model(cfg, sd, device, variant="vidvrd", is_train=False):
    from models import BIG_C_vidor, BIG_C_vidvrd
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    cfg["bias_matrix_path"] = os.path.join(d, "bias.npy")
    if "EntiNameEmb" in sd:
        np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
        cfg["EntiNameEmb_path"] = os.path.join(d, "emb.npy")
    model = (BIG_C_vidvrd if variant == "vidvrd" else BIG_C_vidor)(cfg, is_train=is_train)
    model.load_state_dict(sd)
    return model.to(device)


def init_dist(args):
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, "WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BIGC_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return rank, world, device, dist


def collect_families(lib, steps):
    out = {}
    _lib.check(lib.bigc_prof_collect())
    for f in range(lib.bigc_prof_family_count()):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        lib.bigc_prof_get(f, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        if n.value:
            out[lib.bigc_prof_family_name(f).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out


def timed_with_events(lib, fn, warmup, steps, repeats=3):
    """(wall seconds un-instrumented, wall seconds instrumented, per-family HIP-event times) of `steps` calls of fn.  The extra legs time
    a few steps of a few milliseconds each on a box shared with other jobs: the un-instrumented loop runs `repeats` times and the MEDIAN
    repeat is reported (all of them are kept in timed_with_events.last_repeats_ms for the leg to print)."""
    for _ in range(warmup):
        fn()
    els = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        els.append(time.perf_counter() - t0)
    el = sorted(els)[len(els) // 2]
    timed_with_events.last_repeats_ms = [round(1e3 * e / steps, 3) for e in els]
    lib.bigc_prof_reset()
    lib.bigc_prof_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    el_prof = time.perf_counter() - t0
    lib.bigc_prof_enable(0)
    return el, el_prof, collect_families(lib, steps)


def gemm_roofline(fam, steps, step_ms=None):
    """roofline object of the GEMM family from its HIP-event totals (None when the family did not run).  The family's launches
    carry their algorithmic flops AND bytes (A + W + C (+ residual) once each, csrc/linear.hip); the bound is whichever of
    flops / MFMA peak and bytes / HBM peak takes longer -- MFMA for the K >= 512 GEMMs of BIG-C, HBM for the d = 128 GEMMs of
    the grounding scorer (32 flop per byte)."""
    lin = fam.get("linear_f32")
    if not lin:
        return None
    sec = lin["ms"] * 1e-3
    peak_f = F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3
    t_mfma, t_hbm = lin["flops"] / (peak_f * 1e12), lin["bytes"] / (HBM_PEAK_GBS * 1e9)
    if t_hbm > t_mfma:
        achieved = lin["bytes"] / sec / 1e9
        r = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                 gbytes_per_step=round(lin["bytes"] / steps / 1e9, 3), tflops=round(lin["flops"] / sec / 1e12, 2))
    else:
        achieved = lin["flops"] / sec / 1e12
        r = dict(bound="mfma", achieved=round(achieved, 2), peak=round(peak_f, 1), unit="TFLOP/s", frac=round(achieved / peak_f, 4))
    r.update(launches_per_step=lin["launches"] // steps, avg_launch_us=round(1e3 * lin["ms"] / lin["launches"], 2),
             gflop_per_step=round(lin["flops"] / steps / 1e9, 1))
    if step_ms:
        r["share_of_step"] = round(lin["ms"] / steps / step_ms, 3)
    return r


def family_rooflines(fam, steps, step_ms, mfma_peak_tflops=None):
    """Every kernel family of a leg against the roof that bounds it: the longer of flops / MFMA peak and bytes / HBM peak decides the bound
    (families whose launches carry neither -- index tables, sorts -- are listed with their time only).  -> (dict per family, share of
    the instrumented step that has a roofline, total GFLOP per step, total algorithmic GB per step)"""
    peak_f = (F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3) if mfma_peak_tflops is None else mfma_peak_tflops
    out, covered, flops_all, bytes_all = {}, 0.0, 0.0, 0.0
    for name, v in fam.items():
        sec = v["ms"] * 1e-3
        r = dict(ms_per_step=round(v["ms"] / steps, 4), launches_per_step=v["launches"] // steps)
        if sec > 0 and (v["flops"] > 0 or v["bytes"] > 0):
            t_mfma, t_hbm = v["flops"] / (peak_f * 1e12), v["bytes"] / (HBM_PEAK_GBS * 1e9)
            if t_hbm >= t_mfma:
                a = v["bytes"] / sec / 1e9
                r.update(bound="hbm", achieved=round(a, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(a / HBM_PEAK_GBS, 4))
            else:
                a = v["flops"] / sec / 1e12
                r.update(bound="mfma", achieved=round(a, 2), peak=round(peak_f, 1), unit="TFLOP/s", frac=round(a / peak_f, 4))
            r.update(gflop_per_step=round(v["flops"] / steps / 1e9, 2), gbytes_per_step=round(v["bytes"] / steps / 1e9, 4))
            covered += v["ms"]
            flops_all += v["flops"]
            bytes_all += v["bytes"]
        out[name] = r
    share = covered / steps / step_ms if step_ms else None
    return out, (round(share, 3) if share is not None else None), flops_all / steps / 1e9, bytes_all / steps / 1e9


# ---------------------------------------------------------------------------------------------------------------------
# --mode train: the training step as its own benchmark line
# ---------------------------------------------------------------------------------------------------------------------

def make_train_job(args, rank, device, nvid):
    from bigc_hip.synthetic import synthetic_gt_graph
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=3.0)
    model = build_model(cfg, sd, device, "vidvrd", is_train=True)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    props = [synthetic_proposal(5000 + rank * nvid + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
             for v in range(nvid)]
    gts = [synthetic_gt_graph(p, 6000 + rank * nvid + v, n_gt_trajs=8, n_preds=10, num_pred_cats=cfg["num_pred_cats"]) for v, p in enumerate(props)]
    props = [p.to(device) for p in props]            # GT stays on the host: alignment + Hungarian run there, as in the reference
    gt_duras = [g.traj_durations.clone() for g in gts]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)     # the same update as the reference's Adam, one kernel
    pairs = sum(p.num_proposals * (p.num_proposals - 1) for p in props)
    return model, props, gts, gt_duras, opt, pairs


def train_step_fn(model, props, gts, gt_duras, opt, dist):
    from bigc_hip.shard import allreduce_gradients, clip_grad_norm_
    from bigc_hip.train_step import mark
    params = list(model.parameters())

    def step():
        for g, du in zip(gts, gt_duras):
            g.traj_durations = du.clone()            # the forward decrements them in place (reference quirk, model_0v10.py:567)
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss.mean().backward()
        mark("autograd glue after the backward")
        flat = allreduce_gradients(params, dist)
        clip_grad_norm_(params, 5, flat=flat)
        opt.step()
        mark("all-reduce + clip + Adam launches")
        # the loss stays on the device: like the reference's loop (tools/train_vidvrd.py:160-164, which reads it every 10th
        # iteration) the step does not wait for the GPU, so the next step's packing overlaps this step's backward
        return loss.detach()
    return step


def train_main(args):
    """`--mode train`: the training step of BASELINE configs[4] (BIG-C DP training) on synthetic videos:
    zero_grad, HIP forward + matching (scipy's assignment algorithm as a device kernel), HIP backward, one flat-gradient all-reduce over RCCL (N>1),
    clip_grad_norm_(5), Adam -- the loop body of tools/train_vidvrd.py:151-164. Not the headline metric."""
    from bigc_hip.shard import allreduce_gradients, max_over_ranks
    rank, world, device, dist = init_dist(args)
    nvid = args.train_videos
    model, props, gts, gt_duras, opt, pairs = make_train_job(args, rank, device, nvid)
    step = train_step_fn(model, props, gts, gt_duras, opt, dist)
    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_this_rank = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_this_rank, dist, device)
    from bigc_hip.shard import flat_gradient_view, rank_report, time_allreduce
    ranks = rank_report(dist, device, elapsed_this_rank, args.steps)
    # the flat-gradient all-reduce alone (every rank takes part): microseconds, algorithmic and bus GB/s over the 111 MB buffer
    allreduce_alone = time_allreduce(flat_gradient_view([p for p in model.parameters() if p.grad is not None]), dist)
    families = {}
    if not args.no_prof:        # per-family kernel time from a few extra steps: the event pairs around every launch slow the host down
        n_prof = min(args.steps, 5)
        lib.bigc_prof_reset()
        lib.bigc_prof_enable(1)
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        lib.bigc_prof_enable(0)
        families = {k: round(v["ms"] / n_prof, 3) for k, v in collect_families(lib, n_prof).items()}
    if rank == 0:
        print(json.dumps({"metric": "train_tracklet_pairs_per_sec", "value": round(pairs * world * args.steps / elapsed, 1), "unit": "pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BIG-C model_0v10 training step (exp2 dims), %d synthetic videos x N=%d per GPU per step, "
                                                 "forward + Hungarian + backward + flat-gradient all-reduce + clip + Adam" % (nvid, args.n_trajs),
                                     "parallelism": "dp%d" % world}, "last_loss": round(float(last), 4),
                          "allreduce": {"backend": "nccl (RCCL)" if dist is not None else None,
                                        "flat_buffer_reduced_in_place": allreduce_gradients.last_in_place,
                                        "bytes": 4 * sum(p.numel() for p in model.parameters() if p.requires_grad),
                                        "alone": allreduce_alone},
                          "ranks": ranks, "kernel_families_ms_per_step": families}))
    if dist is not None:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# extra legs of the eval line (N=1, rank 0, outside the timed region)
# ---------------------------------------------------------------------------------------------------------------------

def device_proposal(seed, n_trajs, feat_width, num_enti_cats, len_range, start_range, device):
    """synthetic_proposal with the (large) feature block drawn on the GPU: same structure recipe (numpy-seeded lengths,
    starts, categories, scores, boxes), features ~ N(0,1) from a seeded device generator.  For the legs that have no
    oracle comparison: 8 VidOR-sized videos are 4 GB of features, minutes of host RNG."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(len_range[0], len_range[1], size=n_trajs)
    starts = rs.randint(start_range[0], start_range[1], size=n_trajs)
    duras = np.stack([starts, starts + lens - 1], axis=1).astype(np.int64)
    cat_ids = rs.randint(1, num_enti_cats, size=n_trajs).astype(np.int64)
    scores = np.sort(rs.uniform(0, 1, size=n_trajs).astype(np.float32))[::-1].copy()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    total = int(lens.sum())
    xy = rs.uniform(0, 600, size=(total, 2))
    wh = rs.uniform(10, 110, size=(total, 2))
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], axis=1).astype(np.float32)).to(device)
    feats = torch.randn((total, feat_width), generator=gen, device=device, dtype=torch.float32)
    off = np.concatenate([[0], np.cumsum(lens)])
    return TrajProposal.from_blocks("synth_%d" % seed, int((starts + lens).max()), (1280, 720), cat_ids, scores, duras, lens, boxes, feats)


def vidor_grounding_legs(args, lib, device, want_vidor, want_grounding):
    """BASELINE configs[3]: VidOR-sized classification (model_0v7 at the exp5 dims, N=180 tracklets, L in [15,1000),
    topk=3, SURVEY.md 8d "config 4") and configs[2]/[3] second stage: the grounding scorer (config G) on the triplet
    queries the classification stage produced for the same videos (rt_triplets_topk = 200 per video)."""
    from models import DEBUG
    from bigc_hip.synthetic import grd_state_shapes
    cfg = dict(configs.CONFIG_O)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidor"), 11, att_gain=8.0)
    big = build_model(cfg, sd, device, "vidor")
    width = cfg["dim_feat"] + cfg["dim_clsme"]
    nv, ntr, max_len = args.vidor_videos, 180, 1000
    # the first `--vidor-oracle-videos` videos are generated on the HOST (numpy-seeded, SURVEY.md 8d recipe) so that the CPU oracle can
    # score the very same inputs; the others draw their (large) feature block on the device
    n_host = min(args.vidor_oracle_videos, nv) if want_vidor else 0
    host_props = [synthetic_proposal(4321 + v, n_trajs=ntr, feat_width=width, num_enti_cats=cfg["num_enti_cats"], len_range=(15, max_len),
                                     start_range=(0, max_len)) for v in range(n_host)]
    import copy
    props = [copy.deepcopy(p).to(device) for p in host_props] + \
            [device_proposal(4321 + v, ntr, width, cfg["num_enti_cats"], (15, max_len), (0, max_len), device) for v in range(n_host, nv)]
    video_len = [p.video_len for p in props]
    packed = PackedBatch(props, device)
    pairs, rows = packed.n_pairs(), packed.n_rows
    props, packed.proposals = None, None
    out_holder = {}

    def cls_step():
        out_holder["o"] = big.forward_packed(packed, 3)
        big.counts(out_holder["o"])

    steps = 3
    el, el_prof, fam = timed_with_events(lib, cls_step, 1, steps)
    legs = {}
    if want_vidor:
        legs["vidor"] = dict(value=round(pairs * steps / el, 1), unit="pairs/s", ms_per_step=round(1e3 * el / steps, 3),
                             ms_per_video=round(1e3 * el / steps / nv, 3), steps=steps,
                             workload="VidOR classification stage (BASELINE configs[3]): BIG-C model_0v7 eval forward incl. construct_triplet, "
                                      "exp5 dims (RoI 1024 + classeme 300, 6 enc / 4 dec), %d synthetic videos x N=%d, L in [15,%d), topk=3, "
                                      "%d rows, %d pairs per step" % (nv, ntr, max_len, rows, pairs),
                             roofline=gemm_roofline(fam, steps, 1e3 * el_prof / steps),
                             kernel_families_ms_per_step={k: round(v["ms"] / steps, 3) for k, v in fam.items()})
    if want_vidor and n_host:
        # the host-seeded videos through the CPU oracle (models/model_0v7.py:434-513 restated), outside the timing: rows identical, logits 1e-4
        from oracle import bigc_oracle as O
        Q, P = cfg["num_querys"], cfg["num_pred_cats"]
        o = out_holder["o"]
        hip_logits = o["pred_logits"][:n_host * Q].cpu().numpy().reshape(n_host, Q, P)
        hip_res = big.unpack_results(nv, list(range(nv)), o)[:n_host]
        torch.set_num_threads(usable_cores())
        rows_same, rows_all, max_rel, t0 = 0, 0, 0.0, time.perf_counter()
        with torch.no_grad():
            for v, p in enumerate(host_props):
                oo = O.encode2decode(sd, cfg, p, "vidor")
                want = O.construct_triplet(p, oo["pred_logits"], oo["att"], 3)
                got = hip_res[v]
                n_w = 0 if want is None else int(want[0].shape[0])
                rows_all += n_w
                if want is not None and got is not None and got[0].shape[0] == n_w:
                    rows_same += int(((got[0].cpu() == want[0]).all(1) & (got[2].cpu() == want[2]).all(1) & (got[3].cpu() == want[3])).sum())
                lg = oo["pred_logits"].numpy()
                max_rel = max(max_rel, float(np.abs(hip_logits[v] - lg).max() / np.abs(lg).max()))
        legs["vidor"]["oracle_parity"] = dict(videos=n_host, identical_triplet_rows="%d of %d" % (rows_same, rows_all), max_logit_rel=max_rel,
                                              oracle_seconds=round(time.perf_counter() - t0, 1),
                                              note="host-seeded videos of this leg (N = 180, L in [15, 1000)) through oracle/bigc_oracle.py, variant vidor")
        host_props = None
    if not want_grounding:
        return legs
    # the grounding leg scores clip sequences of T = ceil(video_len / 8) in [60, 400]: its queries come from the classification
    # stage run (untimed) on videos whose lengths are spread accordingly -- the 16 videos above all have video_len ~ 2000, T ~ 245
    del packed
    g_lens = [480, 800, 1100, 1400, 1800, 2200, 2700, 3200]
    g_props = [device_proposal(5321 + v, ntr, width, cfg["num_enti_cats"], (15, max(30, L // 2)), (0, max(30, L // 2)), device) for v, L in enumerate(g_lens)]
    video_len = [p.video_len for p in g_props]
    nv = len(g_props)
    g_packed = PackedBatch(g_props, device)
    g_props, g_packed.proposals = None, None
    out_holder["o"] = big.forward_packed(g_packed, 3)
    big.counts(out_holder["o"])
    results = big.unpack_results(nv, list(range(nv)), out_holder["o"])
    del g_packed
    gcfg = dict(configs.CONFIG_G)
    gcfg.setdefault("loss_factor", {})
    gsd = seeded_state_dict(grd_state_shapes(gcfg), 12, 4.0)
    d = tempfile.mkdtemp()
    np.save(d + "/ge.npy", gsd["EntiNameEmb"].numpy())
    np.save(d + "/gp.npy", gsd["PredNameEmb"].numpy())
    grd = DEBUG(dict(gcfg, EntiNameEmb_path=d + "/ge.npy", PredNameEmb_path=d + "/gp.npy"), is_train=False)
    grd.load_state_dict(gsd)
    grd = grd.to(device)
    jobs = []
    rs = np.random.RandomState(99)
    for v, r in enumerate(results):
        if r is None:
            continue
        quint, scores, dura = r[0], r[1], r[2]
        if quint.shape[0] > 200:                     # rt_triplets_topk (experiments/exp6/config_rt200.py:14)
            keep = torch.sort(torch.argsort(scores.mean(dim=-1), descending=True)[:200])[0]
            quint, dura = quint[keep], dura[keep]
        T = -(-video_len[v] // 8)
        clip = torch.from_numpy((0.1 * rs.standard_normal((T, gcfg["dim_feat"]))).astype(np.float32)).to(device)
        jobs.append((clip, quint.to(device), dura.to(device), video_len[v]))
    del big, out_holder
    if not jobs:
        return legs
    n_q = sum(int(j[1].shape[0]) for j in jobs)

    def grd_step():
        if args.grd_one_video_per_call:
            for clip, quint, dura, vl in jobs:
                last = grd.forward_raw(clip, quint, dura, vl, 0.9, 0.5, 0.2, 0.8)
        else:
            last = grd.forward_raw_batch(jobs, 0.9, 0.5, 0.2, 0.8)[-1]
        if args.grd_one_video_per_call:
            last["status"].cpu()         # the reference's loop reads every video's result before the next forward
        # batched: results stay on the device; like the headline loop, the timed region waits for the GPU once at its end, so
        # the host's packing of the next batch overlaps this one's kernels

    # ---- parity of THIS leg's workload (outside the timing): the two videos with the fewest query x clip rows through the CPU oracle
    # (oracle/grd_oracle.py: models/grd_model_v5.py:310-373,530-576,667-737 restated) -- the three heads of forward_propagation against the
    # HIP path's, and the final pooled spans / bin probabilities / bin masks of the batched call
    grd_parity = None
    if args.grd_oracle_videos > 0:
        from oracle import grd_oracle as G
        torch.set_num_threads(usable_cores())
        order = sorted(range(len(jobs)), key=lambda i: int(jobs[i][0].shape[0]) * int(jobs[i][1].shape[0]))[:args.grd_oracle_videos]
        outs = grd.forward_raw_batch(jobs, 0.9, 0.5, 0.2, 0.8)
        k = gcfg["num_bins"]
        hm, pm, bm, m_same, m_all, t0 = 0.0, 0.0, 0.0, 0, 0, time.perf_counter()
        shapes = []
        with torch.no_grad():
            for i in order:
                clip, quint, dura, vl = jobs[i]
                n, T = int(quint.shape[0]), int(clip.shape[0])
                shapes.append((n, T))
                one = grd.forward_raw(clip, quint, dura, vl, 0.9, 0.5, 0.2, 0.8, keep_intermediates=True)
                words, info = G.prepare_data(gsd, quint.cpu(), dura.cpu(), vl)
                regrs, conf, cls = G.forward_propagation(gsd, clip.cpu(), words, info)
                for key, want, w in (("regrs", regrs, 2 * k), ("conf", conf, k), ("cls", cls, k)):
                    got = one[key].cpu().numpy().reshape(n, T, w)
                    hm = max(hm, float(np.abs(got - want.numpy()).max() / np.abs(want.numpy()).max()))
                pooled, probs, mask = G.forward(gsd, gcfg, clip.cpu(), (quint.cpu(), dura.cpu(), vl), 0.9, 0.5, 0.2, 0.8)
                o = outs[i]
                pm = max(pm, float((o["pooled_se"].cpu() - pooled).abs().max()))
                bm = max(bm, float((o["bins_probs"].cpu() - probs).abs().max()))
                m_same += int((o["bins_mask"].cpu().bool() == mask.bool()).sum())
                m_all += int(mask.numel())
        grd_parity = dict(videos=len(order), queries_x_clips=["%d x %d" % s for s in shapes], heads_max_rel=hm, pooled_se_max_abs=pm, bins_probs_max_abs=bm,
                          bins_mask_identical="%d of %d" % (m_same, m_all), oracle_seconds=round(time.perf_counter() - t0, 1),
                          note="oracle/grd_oracle.py on videos of this leg: heads relative to max |head output|, spans (normalised to [0, 1]) and probabilities absolute")
        del outs
    gsteps = 5
    el, el_prof, fam = timed_with_events(lib, grd_step, 2, gsteps)
    by_fam, share, gflop_step, gbyte_step = family_rooflines(fam, gsteps, 1e3 * el_prof / gsteps)
    clip_bytes = sum(int(j[0].numel()) * 4 for j in jobs)
    weight_bytes = 4 * sum(int(v.numel()) for v in gsd.values())
    out_bytes = n_q * (gcfg["num_bins"] + 1) * (2 * 4 + 4 + 1)
    # ~23 GFLOP per video of 200 queries x 120 clips; this leg's videos have other n x T, so the figure is scaled by sum n T
    nt_sum = sum(int(j[1].shape[0]) * int(j[0].shape[0]) for j in jobs)
    grd_step = dict(gflop_per_step_counted=round(gflop_step, 1), gflop_per_step_survey_8d_scaled=round(23.0 * nt_sum / (200 * 120), 1),
                    algorithmic_gbytes_per_step_all_launches=round(gbyte_step, 3),
                    compulsory_bytes=dict(clip_features=clip_bytes, weights=weight_bytes, outputs=out_bytes, total=clip_bytes + weight_bytes + out_bytes),
                    roofline_share_of_step=share,
                    note="every family of the leg with its own bound (rooflines_by_family): the d = 128 GEMMs, depth-wise stages, LayerNorm and attention are "
                         "HBM-bound on the INTERMEDIATES they exchange (32 flop per byte at d = 128); against the compulsory bytes alone (clip features + weights + outputs) "
                         "the step would take %.1f us at 8 TB/s" % (1e6 * (clip_bytes + weight_bytes + out_bytes) / (HBM_PEAK_GBS * 1e9)))
    legs["grounding"] = dict(value=round(n_q * gsteps / el, 1), unit="triplet-queries/s", ms_per_step=round(1e3 * el / gsteps, 3), rooflines_by_family=by_fam, step=grd_step,
                             ms_per_video=round(1e3 * el / gsteps / len(jobs), 3), steps=gsteps, ms_per_step_of_each_repeat=timed_with_events.last_repeats_ms,
                             workload="grounding stage (BASELINE configs[2]/[3]): grd_model_v5 DEBUG eval forward incl. temporal pooling + NMS, "
                                      "config G, %d videos, %d queries (<=200 per video), T = ceil(video_len/8) in [%d,%d]"
                                      % (len(jobs), n_q, min(int(j[0].shape[0]) for j in jobs), max(int(j[0].shape[0]) for j in jobs)),
                             roofline=gemm_roofline(fam, gsteps, 1e3 * el_prof / gsteps), oracle_parity=grd_parity,
                             kernel_families_ms_per_step={k: round(v["ms"] / gsteps, 3) for k, v in fam.items()})
    return legs


def fast_mode_leg(args, lib, model, packed, step, ref_out, cfg, base, width, pairs_per_step, device):
    """The opt-in reduced-precision mode (bigc_linear_set_fast: every GEMM operand rounded to one f16, one MFMA per product; the
    "bf16 fast mode" of SURVEY.md section 7 step 10 / BASELINE configs[1]) on the same workload, outside the timed region.  It does
    not meet the 1e-4 logit tolerance by design: its gate is SURVEY.md 8(d)'s |R@50(fast) - R@50(parity)| <= 1e-3, evaluated by the
    `recall` leg on a TRAINED model (seeded random weights recall nothing, 0 = 0)."""
    from bigc_hip import evaluate as E
    lib.bigc_linear_set_fast(1)
    try:
        fa_out = model.forward_packed(packed, args.topk)          # also the warm-up of this mode
        model.counts(fa_out)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fast_steps):
            step()
        torch.cuda.synchronize()
        fa_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_linear_set_fast(0)
    # the literal bf16 arithmetic of BASELINE configs[1] (operands rounded to bf16, v_mfma_f32_32x32x16_bf16; no GEMM -> GEMM chains)
    lib.bigc_linear_set_fast(2)
    try:
        bf_out = model.forward_packed(packed, args.topk)
        model.counts(bf_out)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.fast_steps):
            step()
        torch.cuda.synchronize()
        bf_elapsed = time.perf_counter() - t0
    finally:
        lib.bigc_linear_set_fast(0)
    bf_same = (ref_out["soid"] == bf_out["soid"]).all(1).reshape(-1)
    bf16 = dict(value=round(pairs_per_step * args.fast_steps / bf_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * bf_elapsed / args.fast_steps, 3),
                arithmetic="GEMM operands rounded to one bf16 (8-bit mantissa), one v_mfma_f32_32x32x16_bf16 per product, f32 accumulate; every GEMM on the "
                           "f32-operand kernel (the split form that chains GEMMs carries f16 heads); opt-in (BIGC_FAST=bf16 / bigc_linear_set_fast(2))",
                max_logit_diff_rel_vs_parity_mode=float((((ref_out["pred_logits"] - bf_out["pred_logits"]).abs().max(1).values * bf_same).max()
                                                         / ref_out["pred_logits"].abs().max()).item()),
                queries_with_other_argmax_entity=int((~bf_same).sum()), gate="recall.bf16_gate_passed")
    same_so = (ref_out["soid"] == fa_out["soid"]).all(1).reshape(-1)
    logit_rel = float((((ref_out["pred_logits"] - fa_out["pred_logits"]).abs().max(1).values * same_so).max() / ref_out["pred_logits"].abs().max()).item())
    nv = min(32, args.videos)
    host = {k: {n: o[n][:nv].cpu() for n in ("quint", "scores", "dura", "n_out")} for k, o in (("parity", ref_out), ("fast", fa_out))}
    preds, common, total = {"parity": {}, "fast": {}}, 0, 0
    for v in range(nv):
        p = synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
        p.video_name = "synth_%d" % v
        top = {}
        for k, h in host.items():
            n = int(h["n_out"][v])
            res = (h["quint"][v, :n].long(), h["scores"][v, :n], h["dura"][v, :n].long()) if n else None
            preds[k].update(E.to_eval_format(p, res))
            # the 50 best-scored triplets of the video (what Recall@50 looks at), as (quintuple, span) keys
            if n:
                order = torch.argsort(h["scores"][v, :n].mean(-1), descending=True, stable=True)[:50]
                top[k] = {tuple(h["quint"][v, i].tolist()) + tuple(h["dura"][v, i].tolist()) for i in order.tolist()}
            else:
                top[k] = set()
        common += len(top["parity"] & top["fast"])
        total += len(top["parity"])
    # a sensitivity figure on the benchmark model itself (seeded random weights find no relation of a random-predicate ground truth):
    # GT relations taken FROM the parity mode's output -- 12 of each video's 100 best-scored relations -- so that the parity
    # mode's R@50 is about one half and every relation the fast mode ranks differently shows.  The GATE is the `recall` leg's.
    rs = np.random.RandomState(77)
    gts2 = {}
    for name, rel in preds["parity"].items():
        best = sorted(rel, key=lambda x: -x["score"])[:100]
        pick = rs.choice(len(best), size=min(12, len(best)), replace=False) if best else []
        gts2[name] = [{k: best[i][k] for k in ("triplet", "duration", "sub_traj", "obj_traj")} for i in sorted(pick)]
    _, r2_p, _ = E.evaluate(gts2, preds["parity"])
    _, r2_f, _ = E.evaluate(gts2, preds["fast"])
    return dict(value=round(pairs_per_step * args.fast_steps / fa_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * fa_elapsed / args.fast_steps, 3),
                steps=args.fast_steps, arithmetic="GEMM operands rounded to one f16 (11-bit mantissa), one v_mfma_f32_32x32x16_f16 per product, f32 accumulate; "
                                                  "attention kernels unchanged; opt-in (BIGC_FAST=1 / bigc_linear_set_fast), never the headline",
                max_logit_diff_rel_vs_parity_mode=logit_rel, queries_with_other_argmax_entity=int((~same_so).sum()),
                top50_triplets_in_common_on_the_benchmark_model="%d of %d" % (common, total), bf16=bf16,
                recall_on_model_derived_gt=dict(gt_relations=sum(len(x) for x in gts2.values()), recall50_parity=r2_p[50], recall50_fast=r2_f[50],
                                                recall100_parity=r2_p[100], recall100_fast=r2_f[100], abs_diff_recall50=abs(r2_p[50] - r2_f[50]),
                                                note="GT = 12 of the parity mode's 100 best relations per video: a sensitivity figure; the gate is in `recall`"))


def batch512_leg(args, lib, device):
    """The same eval step on twice the headline's videos per step (2 x --videos, default 512): from this size on the decoder's GEMM
    operands are written in the split form by their producers and the whole decoder runs on the pre-split-operand GEMM (automatic,
    csrc/model.hip; profiles/HISTORY_rounds1-4.md section 5 round 3).  Not the headline -- the default step stays at the survey's 256-video set -- but in
    the same line, so that the batch-size sensitivity is the driver's measurement and not a number in a document."""
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    nv = 2 * args.videos
    t0 = time.time()
    props = [synthetic_proposal(70000 + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(device) for v in range(nv)]
    packed = PackedBatch(props, device)
    props = None
    packed.proposals = None
    gen_s = time.time() - t0
    model = build_model(cfg, sd, device)
    pairs = packed.n_pairs()

    def step():
        return model.counts(model.forward_packed(packed, args.topk))

    steps = 5
    el, el_prof, fam = timed_with_events(lib, step, 3, steps)
    roof = gemm_roofline(fam, steps, 1e3 * el_prof / steps)
    out = dict(value=round(pairs * steps / el, 1), unit="pairs/s", videos_per_step=nv, ms_per_step=round(1e3 * el / steps, 3), steps=steps,
               roofline=roof, setup_s=round(gen_s, 1),
               note="same workload and model as the headline at twice the videos per step; decoder GEMMs on pre-split operands (automatic from 512 videos)")
    model._engine = None
    del model, packed
    torch.cuda.empty_cache()
    return out


def recall_leg(args, lib, device):
    """BASELINE's metric, second half ("+ Recall@50 parity"), on relations that are actually FOUND: the exp2-dims model is trained for
    `--recall-steps` iterations of the HIP training step (the reference's loop body, tools/train_vidvrd.py:151-164) on the learnable
    synthetic relation set (bigc_hip/recall_set.py: marked actor tracklets, 12 GT relations per video as SURVEY.md 8d), then
    `--recall-videos` HELD-OUT videos are scored by the HIP eval path (default mode and the opt-in reduced-precision mode) and by the
    CPU oracle with the same trained weights, through the evaluator of VidVRDhelperEvalAPIs/visual_relation_detection.py:64-120
    (bigc_hip/evaluate.py).  Gates: |R@50(HIP) - R@50(oracle)| = 0 in parity mode, <= 1e-3 in the reduced-precision mode."""
    import copy
    from bigc_hip import evaluate as E, recall_set as R
    from oracle import bigc_oracle as O
    cfg = dict(configs.CONFIG_V)
    t0 = time.perf_counter()
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 11, att_gain=1.0)
    torch.manual_seed(1234)
    trainee = build_model(cfg, sd, device, "vidvrd", is_train=True)
    # a HARD version of the set (round 4): weaker actor markers and 30 % label noise in the ground truth (training and held-out), so that a
    # trained model's Recall@50 sits well inside (0, 1) -- about 1 - noise -- where a change of arithmetic can move it; without them every
    # mode scored 1.0 and the gate could not tell them apart (benchmarks/probes/recall_probe.py: settings tried, incl. a motion-dependent
    # predicate whose training outcome varied too much from run to run to gate on)
    hard = dict(n_actors=4, marker=1.5, label_noise=0.3)
    data = R.relational_set(cfg, 20000, 64, n_trajs=args.n_trajs, **hard)
    for p, _ in data:
        p.to(device)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    last_loss = R.train(trainee, data, args.recall_steps, batch=16, lr=1e-4)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    sd_t = {k: v.detach().cpu() for k, v in trainee.state_dict().items()}
    del trainee, data
    torch.cuda.empty_cache()
    model = build_model(cfg, sd_t, device, "vidvrd", is_train=False)
    held = R.relational_set(cfg, 30000, args.recall_videos, n_trajs=args.n_trajs, **hard)
    gts = R.ground_truth(held)
    props = [p for p, _ in held]
    dev_props = [copy.deepcopy(p).to(device) for p in props]
    rel = {"hip": R.relations(props, model(dev_props, topk=args.topk))}
    if not EXACT_F32:
        lib.bigc_linear_set_fast(1)
        try:
            rel["hip_fast"] = R.relations(props, model(dev_props, topk=args.topk))
            lib.bigc_linear_set_fast(2)
            rel["hip_bf16"] = R.relations(props, model(dev_props, topk=args.topk))
        finally:
            lib.bigc_linear_set_fast(0)
    torch.set_num_threads(usable_cores())
    t0 = time.perf_counter()
    with torch.no_grad():
        rel["oracle"] = R.relations(props, O.forward_test(sd_t, cfg, props, args.topk))
    oracle_s = time.perf_counter() - t0
    res = {k: E.evaluate(gts, v) for k, v in rel.items()}          # (mean AP, recall {50, 100}, tagging precision)
    out = dict(trained="exp2 dims, %d iterations x 16 videos (N=%d) on the HIP training step, Adam lr 1e-4, dropout on; last loss %.3f; %.1f s (+ %.1f s of data)"
                       % (args.recall_steps, args.n_trajs, last_loss, train_s, gen_s),
               held_out_videos=len(held), gt_relations=sum(len(x) for x in gts.values()),
               recall50_oracle=res["oracle"][1][50], recall50_hip=res["hip"][1][50], recall100_oracle=res["oracle"][1][100], recall100_hip=res["hip"][1][100],
               mean_ap_oracle=res["oracle"][0], mean_ap_hip=res["hip"][0],
               relations_oracle=sum(len(x) for x in rel["oracle"].values()), relations_hip=sum(len(x) for x in rel["hip"].values()),
               abs_diff_recall50_hip_vs_oracle=abs(res["hip"][1][50] - res["oracle"][1][50]),
               relation_set="marked actors (marker 1.5), predicate = f(actor pair), 30 % of the ground-truth predicates replaced by random classes",
               recall50_inside_the_discriminating_band_0p5_0p9=bool(0.5 <= res["oracle"][1][50] <= 0.9),
               parity_gate="|R@50(HIP) - R@50(oracle)| = 0 at R@50 > 0 (SURVEY.md 8d)",
               parity_gate_passed=bool(res["oracle"][1][50] > 0 and res["hip"][1][50] == res["oracle"][1][50] and res["hip"][1][100] == res["oracle"][1][100]),
               oracle_seconds=round(oracle_s, 1))
    if "hip_fast" in res:
        out.update(recall50_hip_fast=res["hip_fast"][1][50], recall100_hip_fast=res["hip_fast"][1][100], mean_ap_hip_fast=res["hip_fast"][0],
                   abs_diff_recall50_fast_vs_parity_mode=abs(res["hip_fast"][1][50] - res["hip"][1][50]),
                   fast_gate="|R@50(reduced precision) - R@50(parity mode)| <= 1e-3 (SURVEY.md 8d)",
                   fast_gate_passed=bool(abs(res["hip_fast"][1][50] - res["hip"][1][50]) <= 1e-3))
    if "hip_bf16" in res:
        out.update(recall50_hip_bf16=res["hip_bf16"][1][50], recall100_hip_bf16=res["hip_bf16"][1][100], mean_ap_hip_bf16=res["hip_bf16"][0],
                   abs_diff_recall50_bf16_vs_parity_mode=abs(res["hip_bf16"][1][50] - res["hip"][1][50]),
                   bf16_gate_passed=bool(abs(res["hip_bf16"][1][50] - res["hip"][1][50]) <= 1e-3))
    model._engine = None
    del model
    torch.cuda.empty_cache()
    return out


def train_pmc_traffic(args, nvid):
    """HBM bytes per GEMM launch and per whole training step from the committed rocprofv3 --pmc passes of `bench.py --mode train
    --train-videos nvid` (profiles/pmc_traffic.json, section "train"; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) -- only when they
    were measured on these kernel sources and this workload; else (None, reason)"""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc_path) or args.n_trajs != 64:
        return None, "no committed counter pass for this workload"
    pmc = json.load(open(pmc_path))
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from pmc_summary import csrc_sha16
    if pmc.get("csrc_sha16") != csrc_sha16():
        return None, "profiles/pmc_traffic.json was measured on other kernel sources (csrc fingerprint differs): not reported"
    t = (pmc.get("train") or {}).get("videos_%d" % nvid)
    if not t or "error" in t:
        return None, "no training pass in profiles/pmc_traffic.json"
    return t, "not measured in this run: profiles/pmc_traffic.json (%s), rocprofv3 --pmc passes of `bench.py --mode train --train-videos %d` on these kernel sources" % (pmc.get("tag", ""), nvid)


def train_leg(args, lib, device):
    """BASELINE configs[4] on one GPU: the BIG-C training step at 8 and 64 videos per step"""
    out = {}
    for nvid in (8, 64):
        model, props, gts, gt_duras, opt, pairs = make_train_job(args, 0, device, nvid)
        step = train_step_fn(model, props, gts, gt_duras, opt, None)
        steps = 8
        el, el_prof, fam = timed_with_events(lib, step, 4, steps)
        repeats_ms = timed_with_events.last_repeats_ms
        kern_ms = sum(v["ms"] for v in fam.values()) / steps       # the library's kernels (HIP events); torch's Adam / clip kernels come on top
        # roofline of the step's GEMMs: forward + dX = dY W (family linear_f32) and dW = dY^T X (family linear_dw), both on the split-operand
        # f16 MFMAs (three per product): algorithmic flops / HIP-event time against the dense f16 peak / 3
        peak_f = F32_MFMA_PEAK_TFLOPS if EXACT_F32 else F16_MFMA_PEAK_TFLOPS / 3
        roof = gemm_roofline(fam, steps, 1e3 * el_prof / steps)
        dw = fam.get("linear_dw")
        if roof is not None and dw:
            lin = fam["linear_f32"]
            both_fl, both_ms = lin["flops"] + dw["flops"], lin["ms"] + dw["ms"]
            roof = dict(bound="mfma", achieved=round(both_fl / both_ms / 1e9, 2), peak=round(peak_f, 1), unit="TFLOP/s",
                        frac=round(both_fl / both_ms / 1e9 / peak_f, 4), kernel="forward + dX + dW GEMMs of the training step (split-operand f16 MFMAs)",
                        forward_and_dx=dict(tflops=roof["achieved"] if roof["unit"] == "TFLOP/s" else roof.get("tflops"), frac=round(lin["flops"] / lin["ms"] / 1e9 / peak_f, 4),
                                            launches_per_step=lin["launches"] // steps, gflop_per_step=round(lin["flops"] / steps / 1e9, 1),
                                            ms_per_step=round(lin["ms"] / steps, 3)),
                        dw=dict(tflops=round(dw["flops"] / dw["ms"] / 1e9, 2), frac=round(dw["flops"] / dw["ms"] / 1e9 / peak_f, 4),
                                launches_per_step=dw["launches"] // steps, gflop_per_step=round(dw["flops"] / steps / 1e9, 1), ms_per_step=round(dw["ms"] / steps, 3)),
                        share_of_step=round(both_ms / steps / (1e3 * el_prof / steps), 3), traffic=None,
                        measured="HIP events around each launch over %d instrumented steps" % steps)
            tpmc, tsrc = train_pmc_traffic(args, nvid)
            roof["traffic_source"] = tsrc
            if tpmc:
                roof["traffic"] = round(tpmc["gemm_hbm_bytes_per_launch"])
                roof["traffic_launches_per_step"] = tpmc["gemm_launches_per_step"]
                roof["algorithmic_bytes_per_launch"] = round((lin["bytes"] + dw["bytes"]) / max(1, lin["launches"] + dw["launches"]))
                roof["step_traffic_all_kernels"] = dict(hbm_bytes=tpmc["whole_step_fetch_bytes_all_kernels"] + tpmc["whole_step_write_bytes_all_kernels"])
        by_fam, share, gflop_step, _ = family_rooflines(fam, steps, 1e3 * el_prof / steps)
        out["videos_%d" % nvid] = dict(value=round(pairs * steps / el, 1), unit="pairs/s", ms_per_step=round(1e3 * el / steps, 3), steps=steps, ms_per_step_of_each_repeat=repeats_ms,
                                       kernel_ms_per_step=round(kern_ms, 3), kernel_share_of_step=round(kern_ms / (1e3 * el / steps), 3),
                                       instrumented_ms_per_step=round(1e3 * el_prof / steps, 3), roofline=roof,
                                       rooflines_by_family={k: v for k, v in by_fam.items() if "bound" in v}, roofline_share_of_step=share,
                                       kernel_families_ms_per_step={k: round(v["ms"] / steps, 3) for k, v in fam.items()})
        del model, props, gts, opt
        torch.cuda.empty_cache()
    out["workload"] = ("BIG-C model_0v10 training step (exp2 dims, dropout on), synthetic videos x N=%d: zero_grad + forward + matching on the device (GT alignment + scipy's assignment algorithm, no host round trip) + "
                       "backward + clip_grad_norm_ + Adam (tools/train_vidvrd.py:151-164)" % args.n_trajs)
    return out


# ---------------------------------------------------------------------------------------------------------------------

def main():
    args = ARGS
    if args.mode == "train":
        return train_main(args)
    rank, world, device, dist = init_dist(args)

    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=8.0)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    # each rank owns its own shard of synthetic videos (weak scaling): seeds 1234 + rank*videos + v
    base = 1234 + rank * args.videos
    t0 = time.time()
    # each video is moved to the GPU as soon as it is generated: the host never holds more than one video of
    # features (the 256-video set is 6.7 GB; 8 ranks of it would not fit in host memory)
    props = [synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"]).to(device)
             for v in range(args.videos)]
    torch.cuda.synchronize()
    n_cpu = min(args.cpu_videos, args.videos) if (rank == 0 and world == 1) else 0
    gen_s = time.time() - t0
    model = build_model(cfg, sd, device)
    lib = _lib.load()

    # ---- the step that is timed: the drop-in call itself, model(proposal_list, topk) (models/model_0v10.py:359-388) on device-resident
    # TrajProposals -- everything the product does per batch is inside: the ragged batch's tables and boxes (the feature rows are read in
    # place through a row-address table, csrc/linear_split.hip / frontend.hip), the forward, construct_triplet, the compaction to the
    # reference's return types, the one D2H copy (row offsets + range flag) and the per-video result tuples
    def api_step():
        return model(props, topk=args.topk)

    # ... and the same forward on a batch that stays packed in HBM (a resident PackedBatch, the engine-level entry point): what a
    # loop over resident data pays per pass; reported as `resident_packed_batch`, never the headline
    packed = PackedBatch(props, device, "lazy")
    if os.environ.get("BIGC_FEAT_HR8", "1") != "0":
        packed.materialize("hr8", cfg["dim_feat"])      # the resident form: RoI columns as f16 pairs + I3D columns f32, gathered by one launch
    pairs_per_step, rows_per_step = packed.n_pairs(), packed.n_rows

    def packed_step():
        out = model.forward_packed(packed, args.topk)
        return model.counts(out)       # the one D2H copy forward() needs (row offsets + the GEMM range flag) -> host sync

    step = packed_step if args.headline == "packed" else api_step
    for _ in range(args.warmup):
        step()
    # ---- the timed region: K steps, no instrumentation
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    thr0, cpu0 = throttled_periods(), time.process_time()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        last = step()
        stamps.append(time.perf_counter())            # (every step ends with its own D2H copy: the stamps are step boundaries)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # what the host was doing meanwhile (diagnostics, outside `value`): CPU seconds of ALL threads of this process per second of wall
    # clock, and how often the container's CPU quota froze the process during the timed steps (cgroup cpu.stat, see bigc_hip/_lib.py)
    host_diag = {"cpu_threads_busy": round((time.process_time() - cpu0) / max(elapsed, 1e-9), 2), "cpu_quota": usable_cores(),
                 "torch_threads": torch.get_num_threads(), "quota_throttled_periods_during_timed_steps": (throttled_periods() - thr0) if thr0 is not None else None,
                 "load_average_1min": round(os.getloadavg()[0], 1)}
    elapsed_this_rank = elapsed
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # who took part (a collective on every rank): rank, host, device ordinal + UUID, each rank's own step time -- the N > 1 line proves
    # through the job's own RCCL group that it ran one process per distinct GPU
    from bigc_hip.shard import rank_report
    ranks = rank_report(dist, device, elapsed_this_rank, args.steps)
    n_triplets_last = int(sum(r[0].shape[0] for r in last if r is not None)) if args.headline != "packed" else int(last.sum())

    # the other of the two entry points, same K steps (rank 0, outside `value`)
    other_leg, pipelined_leg = None, None
    if rank == 0:
        other = api_step if args.headline == "packed" else packed_step
        for _ in range(max(1, args.warmup)):
            other()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            other()
        torch.cuda.synchronize()
        el_o = time.perf_counter() - t1
        other_leg = dict(value=round(pairs_per_step * args.steps / el_o, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * el_o / args.steps, 3), steps=args.steps)
        if not args.no_prof:
            # ... and its GEMM family with HIP events around every launch, as the headline's `roofline` (the two paths differ in the K = 2048
            # GEMM: in-loop split of rows read in place vs the DMA-only kernel on the stored split form)
            lib.bigc_prof_reset()
            lib.bigc_prof_enable(1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                other()
            torch.cuda.synchronize()
            el_op = time.perf_counter() - t1
            lib.bigc_prof_enable(0)
            other_leg["roofline"] = gemm_roofline(collect_families(lib, args.steps), args.steps, 1e3 * el_op / args.steps)
        # the API in two halves (model.forward_async: the host side of call i + 1 -- tables, launches -- runs while the GPU works on call i),
        # as tools/eval_vidvrd.py loops over its chunks; same kernels, same results; a named extra, never the headline
        pend = None
        for _ in range(2):
            nxt = model.forward_async(props, topk=args.topk)
            if pend is not None:
                pend.result()
            pend = nxt
        pend.result()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pend = None
        for _ in range(args.steps):
            nxt = model.forward_async(props, topk=args.topk)
            if pend is not None:
                pend.result()
            pend = nxt
        last_async = pend.result()
        torch.cuda.synchronize()
        el_p = time.perf_counter() - t1
        pipelined_leg = dict(value=round(pairs_per_step * args.steps / el_p, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * el_p / args.steps, 3), steps=args.steps,
                             results_identical_to_the_synchronous_call=bool(all((a is None) == (b is None) and (a is None or all(torch.equal(x, y) for x, y in zip(a, b)))
                                                                               for a, b in zip(last_async, api_step()))),
                             note="model.forward_async(proposal_list) / .result(): two calls in flight")
        del last_async, pend, nxt
        # host-side cost of the API call, each measured on its own with the GPU idle: the tables of the pack, and the result tuples
        tp = time.perf_counter()
        for _ in range(3):
            PackedBatch(props, device, "lazy")
        torch.cuda.synchronize()
        pack_host_ms = 1e3 * (time.perf_counter() - tp) / 3
        o = model.forward_packed(packed, args.topk)
        model.counts(o)
        tp = time.perf_counter()
        for _ in range(3):
            model.unpack_results(packed.n_vid, list(range(packed.n_vid)), o)
        unpack_host_ms = 1e3 * (time.perf_counter() - tp) / 3
        del o

    # ---- the same K steps once more with a HIP event pair around every kernel launch (on the launch stream, inside the
    # library): per-family launch durations for the roofline.  Kept out of `value` because the ~300 event records per step
    # cost a few % of a step; the instrumented wall time is reported next to it.
    # SURVEY.md 8(d) quotes the MEDIAN step time: the same K steps once more, each one waited for (rank 0, outside `value`)
    per_step = []
    if rank == 0:
        for _ in range(args.steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            per_step.append(time.perf_counter() - t1)
    families = {}
    prof_elapsed = None
    if not args.no_prof:
        lib.bigc_prof_reset()
        lib.bigc_prof_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        prof_elapsed = time.perf_counter() - t0
        lib.bigc_prof_enable(0)
        families = collect_families(lib, args.steps)

    # the outputs that are compared with the oracle: those of the TIMED path (a batch gathered the way model(proposal_list) gathers it)
    ref_out = model.forward_packed(PackedBatch(props, device, "lazy") if args.headline == "api" else packed, args.topk, keep_intermediates=False) if rank == 0 else None
    # the same step with every GEMM on the exact f32 MFMA kernel, outside the timed region, for comparison
    exact_leg = None
    if args.exact_steps > 0 and not EXACT_F32 and rank == 0:
        lib.bigc_linear_set_exact(1)
        ex_out = model.forward_packed(packed, args.topk)          # also the warm-up of this mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.exact_steps):
            packed_step()
        torch.cuda.synchronize()
        ex_elapsed = time.perf_counter() - t0
        lib.bigc_linear_set_exact(0)
        # padded arrays: compare the valid rows of every video
        n_a, n_b = ref_out["n_out"].cpu(), ex_out["n_out"].cpu()
        cap = ref_out["quint"].shape[1]
        valid = (torch.arange(cap)[None, :] < torch.minimum(n_a, n_b)[:, None]).to(device)
        same_rows = ((ref_out["quint"] == ex_out["quint"]).all(-1) & (ref_out["dura"] == ex_out["dura"]).all(-1)) & valid
        same = bool(torch.equal(n_a, n_b)) and int(same_rows.sum()) == int(valid.sum())
        exact_leg = dict(value=round(pairs_per_step * args.exact_steps / ex_elapsed, 1), unit="pairs/s (this GPU)",
                         ms_per_step=round(1e3 * ex_elapsed / args.exact_steps, 3), steps=args.exact_steps,
                         triplet_indices_identical_to_default_mode=bool(same),
                         identical_triplet_rows="%d of %d" % (int(same_rows.sum()), int(n_b.sum())),
                         # a query whose attention arg-max (subject / object entity) is a tie at f32 resolution may pick the
                         # other entity in the other mode and then gathers different features: counted, and left out of the
                         # logit comparison
                         queries_with_other_argmax_entity=int((ref_out["soid"] != ex_out["soid"]).any(1).sum()),
                         max_logit_diff_rel_between_modes=float((((ref_out["pred_logits"] - ex_out["pred_logits"]).abs().max(1).values
                                                                  * (ref_out["soid"] == ex_out["soid"]).all(1).reshape(-1)).max()
                                                                 / ex_out["pred_logits"].abs().max()).item()))
        del ex_out

    # the same call with PLAIN accumulation (bigc_set_accum_policy(0): one f32 accumulator per element on the largest tiles -- rounds 1's default):
    # what the blocked accumulation of the default costs, and what it buys (its distance to float64 is filled in by the oracle_parity loop below)
    plain_leg, plain_logits = None, None
    if args.exact_steps > 0 and not EXACT_F32 and rank == 0 and lib.bigc_get_accum_policy() != 0:
        prev_policy = lib.bigc_set_accum_policy(0)
        try:
            pl_out = model.forward_packed(PackedBatch(props, device, "lazy") if args.headline == "api" else packed, args.topk)     # also the warm-up of this policy
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.exact_steps):
                step()
            torch.cuda.synchronize()
            pl_elapsed = time.perf_counter() - t0
        finally:
            lib.bigc_set_accum_policy(prev_policy)
        n_a, n_b = ref_out["n_out"].cpu(), pl_out["n_out"].cpu()
        cap = ref_out["quint"].shape[1]
        valid = (torch.arange(cap)[None, :] < torch.minimum(n_a, n_b)[:, None]).to(device)
        same_rows_pl = ((ref_out["quint"] == pl_out["quint"]).all(-1) & (ref_out["dura"] == pl_out["dura"]).all(-1)) & valid
        plain_logits = pl_out["pred_logits"].cpu().numpy().reshape(args.videos, cfg["num_querys"], cfg["num_pred_cats"])
        plain_leg = dict(value=round(pairs_per_step * args.exact_steps / pl_elapsed, 1), unit="pairs/s (this GPU)", ms_per_step=round(1e3 * pl_elapsed / args.exact_steps, 3),
                         steps=args.exact_steps, policy="0 = plain: one f32 accumulator per element, 256- / 192-row tiles, fc_feat2enti.0 in place on the DMA kernel",
                         identical_triplet_rows_vs_default="%d of %d" % (int(same_rows_pl.sum()), int(n_b.sum())),
                         queries_with_other_argmax_entity=int((ref_out["soid"] != pl_out["soid"]).any(1).sum()),
                         max_logit_diff_rel_vs_default=float(((ref_out["pred_logits"] - pl_out["pred_logits"]).abs().max() / ref_out["pred_logits"].abs().max()).item()))
        del pl_out

    fast_leg = None
    if args.fast_steps > 0 and not EXACT_F32 and rank == 0 and world == 1 and not lib.bigc_linear_get_fast():
        fast_leg = fast_mode_leg(args, lib, model, packed, packed_step, ref_out, cfg, base, width, pairs_per_step, device)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = pairs_per_step * world * args.steps / elapsed
    roofline = None
    if "linear_f32" in families:
        lin = families["linear_f32"]
        roofline = gemm_roofline(families, args.steps, 1e3 * prof_elapsed / args.steps)
        # HBM bytes per launch cannot be sampled from inside this process: they come from the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/pmc_traffic.json,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if the workload differs
        traffic, traffic_source, step_traffic = None, None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.videos == 256 and args.n_trajs == 64:
            pmc = json.load(open(pmc_path))
            sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
            from pmc_summary import csrc_sha16
            if pmc.get("csrc_sha16") == csrc_sha16():
                traffic = round(pmc["hbm_bytes_per_launch"])
                step_traffic = pmc.get("whole_step_fetch_bytes_all_kernels", 0) + pmc.get("whole_step_write_bytes_all_kernels", 0)
                traffic_source = "not measured in this run: profiles/pmc_traffic.json (%s), rocprofv3 --pmc passes of this command on these kernel sources" % pmc.get("tag", "")
            else:
                traffic_source = "profiles/pmc_traffic.json was measured on other kernel sources (csrc fingerprint differs): not reported"
        if EXACT_F32:
            kernel = "linear_f32_kernel (f32 MFMA 32x32x2)"
        else:
            # every f32 product costs three f16 MFMAs (head x head, head x residual, residual x head): the roof for
            # ALGORITHMIC flops is the dense f16 MFMA peak / 3
            kernel = "split-operand GEMM family (f32 operands as 2 x f16, 3 x v_mfma_f32_32x32x16_f16 per product)"
        # what the matrix pipe of this chip sustains on operands that toggle (a register-only MFMA loop, benchmarks/probes/mfma_peak.hip,
        # committed as profiles/mfma_peak.json): context for `frac`, which stays priced against the guide's 2.5 PFLOP/s
        mp_path = os.path.join(ROOT, "profiles", "mfma_peak.json")
        if os.path.exists(mp_path) and not EXACT_F32:
            mp = json.load(open(mp_path))
            roofline["sustained_mfma_peak"] = dict(tflops_f16_random_operands=mp["random_f16_tflops"], tflops_f16_zero_operands=mp["zeros_tflops"],
                                                   three_mfma_gemm_ceiling=round(mp["random_f16_tflops"] / 3, 1),
                                                   frac_of_that_ceiling=round(roofline["achieved"] / (mp["random_f16_tflops"] / 3), 4),
                                                   source="not measured in this run: profiles/mfma_peak.json (benchmarks/probes/mfma_peak.hip)")
        # SURVEY.md 8(d)'s algorithmic bytes of the STEP: features + boxes read once (6.7 kB per pair) + the weights once per batch
        alg_8d = rows_per_step * (width + 4) * 4 + 4 * sum(int(v.numel()) for v in sd.values())
        roofline.update(kernel=kernel, vs_f32_mfma_peak=round(roofline["achieved"] / F32_MFMA_PEAK_TFLOPS, 3), traffic=traffic,
                        traffic_source=traffic_source, algorithmic_bytes_per_launch=round(lin["bytes"] / lin["launches"]),
                        step_traffic_all_kernels=dict(hbm_bytes=step_traffic, algorithmic_bytes_survey_8d=alg_8d,
                                                      ratio=round(step_traffic / alg_8d, 2) if step_traffic else None,
                                                      note="HBM bytes of every kernel of one step (same PMC passes) against features + boxes + weights read once"),
                        measured="HIP events around each launch, over a repeat of the timed %d steps (instrumented step: %.3f ms)"
                                 % (args.steps, 1e3 * prof_elapsed / args.steps))
    cpu_baseline, oracle_parity = None, None
    if n_cpu:
        from oracle import accuracy as A
        from oracle import bigc_oracle as O
        Q, P = cfg["num_querys"], cfg["num_pred_cats"]
        hip_n = ref_out["n_out"].cpu().numpy()
        hip_quint, hip_dura, hip_qid = ref_out["quint"].cpu().numpy(), ref_out["dura"].cpu().numpy(), ref_out["qid"].cpu().numpy()
        hip_logits = ref_out["pred_logits"].cpu().numpy().reshape(args.videos, Q, P)
        hip_soid = ref_out["soid"].cpu().numpy()
        torch.set_num_threads(usable_cores())
        cpu_pairs, n_done, cpu_s = 0, 0, 0.0
        rows_total, rows_same, vids_same, max_rel, other_argmax = 0, 0, 0, 0.0, 0
        set_hip, set_ref = [], []           # distance to float64 of every sampled video: HIP and the f32 oracle side by side
        row_hip, row_ref, row_hip_vs_ref = [], [], 0.0
        set_plain = []                      # ... and of the plain-accumulation policy's logits
        over, over_worse, over_failed, sd64, row_diffs = [], 0, 0, None, []

        def f64_eval(p):
            """the oracle once more in float64 (the arbiter where two f32 evaluations disagree; outside the CPU timing)"""
            nonlocal sd64
            if sd64 is None:
                sd64 = {k: t.double() for k, t in sd.items()}
            _, p64 = A.to_f64({}, p)
            o64 = O.encode2decode(sd64, cfg, p64)
            return o64, O.construct_triplet(p64, o64["pred_logits"], o64["att"], args.topk)

        def same_rows(a_quint, a_dura, a_qid, n_a, res):
            n_r = 0 if res is None else int(res[0].shape[0])
            if n_a != n_r:
                return False
            return n_r == 0 or bool(((a_quint[:n_r] == res[0].numpy()).all(1) & (a_dura[:n_r] == res[2].numpy()).all(1) & (a_qid[:n_r] == res[3].numpy())).all())

        with torch.no_grad():
            O.forward_test(sd, cfg, [synthetic_proposal(base, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])], args.topk)  # warm-up
            for v in range(n_cpu):                                      # bounded: stop after ~15 s of oracle time
                p = synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])   # the same video, regenerated on the host
                t0 = time.perf_counter()
                o = O.encode2decode(sd, cfg, p)                         # = O.forward_test (models/model_0v10.py:369-388), outputs kept
                res = O.construct_triplet(p, o["pred_logits"], o["att"], args.topk)
                cpu_s += time.perf_counter() - t0
                cpu_pairs += p.num_proposals * (p.num_proposals - 1)
                n_done += 1
                # ---- row-for-row comparison with the HIP output of the same video (outside the CPU timing)
                n_o = 0 if res is None else int(res[0].shape[0])
                n_h = int(hip_n[v])
                rows_total += n_o
                m = min(n_o, n_h)
                if m:
                    eq = ((hip_quint[v, :m] == res[0].numpy()[:m]).all(1) & (hip_dura[v, :m] == res[2].numpy()[:m]).all(1)
                          & (hip_qid[v, :m] == res[3].numpy()[:m]))
                    rows_same += int(eq.sum())
                    vid_same = n_o == n_h and bool(eq.all())
                else:
                    vid_same = n_o == n_h
                vids_same += int(vid_same)
                if not vid_same and len(row_diffs) < 8:
                    # a triplet row differs between two f32 evaluations (a top-k / arg-max decision within round-off of a tie):
                    # which of the two does the float64 oracle side with?
                    _, res64 = f64_eval(p)
                    row_diffs.append(dict(video=v, rows_differing=int(max(n_o, n_h) - (int(eq.sum()) if m else 0)),
                                          hip_rows_equal_the_f64_oracles=same_rows(hip_quint[v], hip_dura[v], hip_qid[v], n_h, res64),
                                          f32_oracle_rows_equal_the_f64_oracles=same_rows(res[0].numpy() if res else None, res[2].numpy() if res else None,
                                                                                           res[3].numpy() if res else None, n_o, res64)))
                so_ref = o["att"].argmax(-1).numpy()                    # (2, Q)
                same_so = (hip_soid[v] == so_ref).all(0)
                other_argmax += int((~same_so).sum())
                lg = o["pred_logits"].numpy()
                if same_so.any():
                    d = float(np.abs(hip_logits[v][same_so] - lg[same_so]).max() / np.abs(lg).max())
                    max_rel = max(max_rel, d)
                    if v < args.f64_videos:
                        # the float64 evaluation of the oracle for EVERY sampled video (outside the CPU timing): max-norm per video, and per
                        # query row (|delta| relative to max |logit| of that query: the stricter reading of "1e-4 rel")
                        if sd64 is None:
                            sd64 = {k: t.double() for k, t in sd.items()}
                        r64 = A.f64_logits(sd, cfg, p, sd64=sd64)
                        h64 = hip_logits[v][same_so].astype(np.float64)
                        set_hip.append(float(np.abs(h64 - r64[same_so]).max() / np.abs(r64).max()))
                        set_ref.append(float(np.abs(lg[same_so].astype(np.float64) - r64[same_so]).max() / np.abs(r64).max()))
                        row_hip.append(float((np.abs(h64 - r64[same_so]).max(1) / np.abs(r64[same_so]).max(1)).max()))
                        row_ref.append(float((np.abs(lg[same_so].astype(np.float64) - r64[same_so]).max(1) / np.abs(r64[same_so]).max(1)).max()))
                        row_hip_vs_ref = max(row_hip_vs_ref, float((np.abs(hip_logits[v][same_so] - lg[same_so]).max(1) / np.abs(lg[same_so]).max(1)).max()))
                        if plain_logits is not None:
                            set_plain.append(float(np.abs(plain_logits[v][same_so].astype(np.float64) - r64[same_so]).max() / np.abs(r64).max()))
                    if d > 1e-4 and len(over) < 12:
                        # two f32 evaluations more than 1e-4 apart: a float64 evaluation of the oracle is the truth, and where HIP is more
                        # than 1e-4 from it, the video's own float32 noise floor decides (oracle/accuracy.py; outside the CPU timing)
                        if sd64 is None:
                            sd64 = {k: t.double() for k, t in sd.items()}
                        a = A.arbitrate(sd, cfg, p, hip_logits[v], lg, sd64=sd64, rows=same_so)
                        a.update(video=v, hip_vs_f32_oracle=d)
                        over.append(a)
                        over_worse += int(a["hip_vs_f64"] > max(1e-4, a["f32_oracle_vs_f64"]))
                        over_failed += int(not a["passed"])
                if cpu_s > args.cpu_seconds:
                    break
        cpu_baseline = dict(value=round(cpu_pairs / cpu_s, 1), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                            sample="%d of the same synthetic videos (N=%d), oracle/bigc_oracle.py encode2decode + construct_triplet "
                                   "(= forward_test), fp32 torch CPU, %.1f s" % (n_done, args.n_trajs, cpu_s))
        oracle_parity = dict(videos=n_done, identical_triplet_rows="%d of %d" % (rows_same, rows_total), videos_fully_identical=vids_same,
                             max_logit_rel=max_rel, queries_with_other_argmax_entity=other_argmax,
                             videos_with_differing_rows=row_diffs,
                             distance_to_float64=(dict(videos=len(set_hip), hip_max=float(np.max(set_hip)), hip_mean=float(np.mean(set_hip)),
                                                       hip_videos_over_1e_4=int((np.asarray(set_hip) > 1e-4).sum()),
                                                       f32_oracle_max=float(np.max(set_ref)), f32_oracle_mean=float(np.mean(set_ref)),
                                                       f32_oracle_videos_over_1e_4=int((np.asarray(set_ref) > 1e-4).sum()),
                                                       hip_farther_than_f32_oracle_and_over_1e_4=int(((np.asarray(set_hip) > 1e-4) & (np.asarray(set_hip) > np.asarray(set_ref))).sum()),
                                                       per_query_row=dict(hip_max=float(np.max(row_hip)), f32_oracle_max=float(np.max(row_ref)), hip_vs_f32_oracle_max=row_hip_vs_ref,
                                                                          note="max over queries of max_c |delta logit| / max_c |logit of that query|"),
                                                       note="max_c,q |logit - float64 logit| / max |float64 logit| per video, float64 = oracle/bigc_oracle.py evaluated in float64")
                                                  if set_hip else None),
                             accumulation_policy=int(lib.bigc_get_accum_policy()),
                             videos_over_1e_4=over, videos_over_1e_4_where_hip_is_farther_from_f64_than_the_f32_oracle=over_worse,
                             videos_outside_1e_4_and_their_f32_noise_floor=over_failed,
                             note="HIP output of the timed workload vs the CPU oracle on the same videos; logits compared on the queries whose "
                                  "attention arg-max (subject, object) agrees, relative to max |logit| of the video; the benchmark model's sharp "
                                  "attention amplifies f32 round-off, so where two f32 evaluations differ by more than 1e-4 a float64 evaluation of "
                                  "the oracle is the truth (`distance_to_float64`: every sampled video).  Round 5: blocked accumulation in every GEMM "
                                  "(accumulation_policy 4) keeps HIP within 1e-4 of float64 on every video; `videos_over_1e_4` lists the videos on which the "
                                  "two F32 evaluations differ by more than that, with the f32 noise-floor diagnostic of oracle/accuracy.py (ratio 1.0)")
    if plain_leg is not None and n_cpu and set_plain:
        plain_leg["distance_to_float64"] = dict(videos=len(set_plain), hip_max=float(np.max(set_plain)), hip_mean=float(np.mean(set_plain)),
                                                hip_videos_over_1e_4=int((np.asarray(set_plain) > 1e-4).sum()))
    del ref_out, plain_logits
    legs = {}
    want = [x for x in args.legs.split(",") if x] if world == 1 else []
    reads_split = bool(packed.feats_hr8 is not None)
    model._engine = None
    del model, packed, props, last
    torch.cuda.empty_cache()
    if args.recall_steps > 0 and world == 1:
        legs["recall"] = recall_leg(args, lib, device)
    if want:
        if "vidor" in want or "grounding" in want:
            legs.update(vidor_grounding_legs(args, lib, device, "vidor" in want, "grounding" in want))
            torch.cuda.empty_cache()
        if "train" in want:
            legs["train"] = train_leg(args, lib, device)
        if "batch512" in want and not EXACT_F32:
            torch.cuda.empty_cache()
            legs["batch_x2"] = batch512_leg(args, lib, device)
    line = {
        "metric": "tracklet_pairs_per_sec", "value": round(value, 1), "unit": "pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "VidVRD-test classification-only (BASELINE configs[1]): BIG-C model_0v10 eval forward incl. "
                               "construct_triplet, exp2 dims (RoI 2048 + I3D 832), %d synthetic videos x N=%d tracklets "
                               "per GPU per step, topk=%d, f32 in / f32 accumulate / f32 out (%s)"
                               % (args.videos, args.n_trajs, args.topk, "GEMMs on the exact f32 MFMA" if EXACT_F32 else
                                  "GEMMs and attention on f16 MFMAs with exactly split f32 operands, f32-level error"),
                   "videos_per_gpu_per_step": args.videos, "tracklets_per_video": args.n_trajs,
                   "rows_per_step": rows_per_step, "pairs_per_gpu_per_step": pairs_per_step, "parallelism": "dp%d" % world},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "oracle_parity": oracle_parity, "exact_f32_gemm_mode": exact_leg, "fast_f16_gemm_mode": fast_leg,
        "plain_accumulation": plain_leg,
        "kernel_families_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in families.items()},
        "videos_per_sec": round(args.videos * world * args.steps / elapsed, 1),
        "ms_per_step_median_each_step_synchronised": round(1e3 * float(np.median(per_step)), 3) if per_step else None,
        "triplets_last_step": n_triplets_last, "setup_s": round(gen_s, 1),
        "timed_steps_ms": [round(1e3 * (b - a), 2) for a, b in zip(stamps[:-1], stamps[1:])],
        "host": host_diag, "ranks": ranks,
        "timed_call": ("model(proposal_list, topk) on device-resident TrajProposals: tables + boxes of the ragged batch (the feature rows are read in place "
                       "through a row-address table, no copy) + forward + construct_triplet + compaction + one D2H copy + per-video result tuples" if args.headline == "api" else
                       "model.forward_packed(resident PackedBatch, topk) + the one D2H copy"),
        "forward_api": (dict(value=round(value, 1), unit="pairs/s", ms_per_step=round(ms_per_step, 3), steps=args.steps, note="= the headline") if args.headline == "api" else other_leg),
        "resident_packed_batch": (other_leg if args.headline == "api" else dict(value=round(value, 1), unit="pairs/s", ms_per_step=round(ms_per_step, 3), steps=args.steps, note="= the headline")),
        "forward_api_pipelined": pipelined_leg,
        "api_host_ms": {"pack_tables_and_launches": round(pack_host_ms, 2), "result_tuples": round(unpack_host_ms, 2),
                        "note": "host side of one model(proposal_list) call at this batch size, each measured alone with the GPU idle (inside the timed call)"},
        "modes": {"BIGC_FEAT_HR8": os.environ.get("BIGC_FEAT_HR8", "1"), "BIGC_PRESPLIT": os.environ.get("BIGC_PRESPLIT", "auto"),
                  "BIGC_CHAINED": os.environ.get("BIGC_CHAINED", "1"), "BIGC_FEAT_IN_PLACE": os.environ.get("BIGC_FEAT_IN_PLACE", "1"),
                  "resident_batch_holds_split_features": reads_split},
    }
    line.update(legs)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
