#!/usr/bin/env python3
"""Benchmark of the BIG classification hot path (BIG-C eval forward incl. construct_triplet).

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): tracklet-pairs/sec = videos/s x N(N-1), whole job.
Workload = BASELINE.json configs[1]: VidVRD-test classification-only, model_0v10 at the exp2
dimensions (PKU RoI 2048-d + I3D 832-d), synthetic videos of N=64 tracklets (SURVEY.md 8d recipe,
video v uses numpy RandomState(1234+v)), numpy-seeded weights.  One step = one pass of the HIP
path over one resident batch of `--videos` videos (default 256) per GPU; inputs are packed in HBM
before the timed region.  N>1: one process per GPU (torch.distributed / RCCL only for the barrier
and the max-over-ranks timing -- videos are independent, no data-path collective), weak scaling.

Also printed in the same JSON line:
  roofline     dominant kernel (linear_f32, the f32-MFMA GEMM family): algorithmic flops of its
               launches / their summed duration measured with HIP events inside the library on the
               launch stream during the timed steps, against the 157.3 TFLOP/s f32-MFMA peak
  cpu_baseline the CPU oracle (oracle/bigc_oracle.py, a torch-CPU restatement of the reference's
               op sequence: kind "port") timed on the host cores, rank 0 at N=1 only, on a bounded
               sample of the same videos
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from bigc_hip import _lib, configs  # noqa: E402
from bigc_hip.packed import PackedBatch  # noqa: E402
from bigc_hip.synthetic import bigc_state_shapes, seeded_state_dict, synthetic_proposal  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0   # same table, "Peak BF16/FP16 MFMA" dense
EXACT_F32 = os.environ.get("BIGC_LINEAR_EXACT", "0") not in ("", "0")     # run the whole benchmark on the exact f32-MFMA GEMM kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--videos", type=int, default=256, help="videos per GPU per step")
    ap.add_argument("--n-trajs", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--cpu-videos", type=int, default=256, help="videos of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--mode", choices=["eval", "train"], default="eval", help="eval = headline metric; train = training-step throughput")
    ap.add_argument("--train-videos", type=int, default=8, help="videos per GPU per training step (reference batch size 8)")
    ap.add_argument("--exact-steps", type=int, default=3, help="extra steps timed with the exact f32-MFMA GEMM kernel, reported beside (0 = skip)")
    return ap.parse_args()


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


def build_model(cfg, sd, device):
    from models import BIG_C_vidvrd
    d = tempfile.mkdtemp()
    cfg = dict(cfg)
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
    cfg["bias_matrix_path"], cfg["EntiNameEmb_path"] = os.path.join(d, "bias.npy"), os.path.join(d, "emb.npy")
    model = BIG_C_vidvrd(cfg, is_train=False)
    model.load_state_dict(sd)
    return model.to(device)


def train_main(args):
    """`--mode train`: the training step of BASELINE configs[4] (BIG-C DP training) on synthetic videos:
    zero_grad, HIP forward + host Hungarian, HIP backward, one flat-gradient all-reduce over RCCL (N>1),
    clip_grad_norm_(5), Adam -- the loop body of tools/train_vidvrd.py:151-164. Not the headline metric."""
    from bigc_hip.shard import allreduce_gradients, max_over_ranks
    from bigc_hip.synthetic import synthetic_gt_graph
    from models import BIG_C_vidvrd
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BIGC_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    cfg = dict(configs.CONFIG_V)
    sd = seeded_state_dict(bigc_state_shapes(cfg, "vidvrd"), 7, att_gain=3.0)
    d = tempfile.mkdtemp()
    np.save(os.path.join(d, "bias.npy"), sd["bias_matrix"].numpy())
    np.save(os.path.join(d, "emb.npy"), sd["EntiNameEmb"].numpy())
    cfg["bias_matrix_path"], cfg["EntiNameEmb_path"] = os.path.join(d, "bias.npy"), os.path.join(d, "emb.npy")
    model = BIG_C_vidvrd(cfg, is_train=True)
    model.load_state_dict(sd)
    model = model.to(device)
    width = cfg["dim_feat"] + cfg["dim_i3d"]
    nvid = args.train_videos
    props = [synthetic_proposal(5000 + rank * nvid + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
             for v in range(nvid)]
    gts = [synthetic_gt_graph(p, 6000 + rank * nvid + v, n_gt_trajs=8, n_preds=10, num_pred_cats=cfg["num_pred_cats"]) for v, p in enumerate(props)]
    props = [p.to(device) for p in props]            # GT stays on the host: alignment + Hungarian run there, as in the reference
    gt_duras = [g.traj_durations.clone() for g in gts]
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    pairs = sum(p.num_proposals * (p.num_proposals - 1) for p in props)

    def step():
        for g, du in zip(gts, gt_duras):
            g.traj_durations = du.clone()            # the forward decrements them in place (reference quirk, model_0v10.py:567)
        opt.zero_grad()
        loss, terms = model(props, gts)
        loss.mean().backward()
        allreduce_gradients(model.parameters(), dist)
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=5, norm_type=2)
        opt.step()
        return float(loss)

    for _ in range(args.warmup):
        step()
    lib = _lib.load()
    if not args.no_prof:
        lib.bigc_prof_reset()
        lib.bigc_prof_enable(1)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, device)
    lib.bigc_prof_enable(0)
    families = {}
    if not args.no_prof:
        _lib.check(lib.bigc_prof_collect())
        for f in range(lib.bigc_prof_family_count()):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            lib.bigc_prof_get(f, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
            if n.value:
                families[lib.bigc_prof_family_name(f).decode()] = round(ms.value / args.steps, 3)
    if rank == 0:
        print(json.dumps({"metric": "train_tracklet_pairs_per_sec", "value": round(pairs * world * args.steps / elapsed, 1), "unit": "pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BIG-C model_0v10 training step (exp2 dims), %d synthetic videos x N=%d per GPU per step, "
                                                 "forward + Hungarian + backward + flat-gradient all-reduce + clip + Adam" % (nvid, args.n_trajs),
                                     "parallelism": "dp%d" % world}, "last_loss": round(last, 4),
                          "kernel_families_ms_per_step": families}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.mode == "train":
        return train_main(args)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("BIGC_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

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
    packed = PackedBatch(props, device)
    props = None
    packed.proposals = None
    n_cpu = min(args.cpu_videos, args.videos) if (rank == 0 and world == 1) else 0
    gen_s = time.time() - t0
    model = build_model(cfg, sd, device)
    lib = _lib.load()
    pairs_per_step = packed.n_pairs()

    def step():
        out = model.forward_packed(packed, args.topk)
        return out["n_out"].cpu()      # the one D2H copy forward() needs (row counts) -> host sync

    for _ in range(args.warmup):
        step()
    # ---- the timed region: K steps, no instrumentation
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K steps once more with a HIP event pair around every kernel launch (on the launch stream, inside the
    # library): per-family launch durations for the roofline.  Kept out of `value` because the ~300 event records per step
    # cost about 4 % of a step; the instrumented wall time is reported next to it.
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
        _lib.check(lib.bigc_prof_collect())
        for f in range(lib.bigc_prof_family_count()):
            ms, n, fl, by = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            lib.bigc_prof_get(f, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
            if n.value:
                families[lib.bigc_prof_family_name(f).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)

    # the same step with every GEMM on the exact f32 MFMA kernel, outside the timed region, for comparison
    exact_leg = None
    if args.exact_steps > 0 and not EXACT_F32 and rank == 0:
        ref_out = model.forward_packed(packed, args.topk)
        lib.bigc_linear_set_exact(1)
        ex_out = model.forward_packed(packed, args.topk)          # also the warm-up of this mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.exact_steps):
            step()
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

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = pairs_per_step * world * args.steps / elapsed
    roofline = None
    if "linear_f32" in families:
        lin = families["linear_f32"]
        achieved = lin["flops"] / (lin["ms"] * 1e-3) / 1e12
        # HBM bytes per launch cannot be sampled from inside this process: they come from the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/pmc_traffic.json,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if the workload differs
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.videos == 256 and args.n_trajs == 64:
            traffic = round(json.load(open(pmc_path))["hbm_bytes_per_launch"])
        if EXACT_F32:
            kernel, peak = "linear_f32_kernel (f32 MFMA 32x32x2)", F32_MFMA_PEAK_TFLOPS
        else:
            # every f32 product costs three f16 MFMAs (head x head, head x residual, residual x head): the roof for
            # ALGORITHMIC flops is the dense f16 MFMA peak / 3
            kernel, peak = "linear_split_kernel (f32 operands split into 2 x f16, 3 x v_mfma_f32_32x32x16_f16 per product)", F16_MFMA_PEAK_TFLOPS / 3
        roofline = dict(bound="mfma", kernel=kernel, achieved=round(achieved, 2),
                        peak=round(peak, 1), unit="TFLOP/s", frac=round(achieved / peak, 4),
                        vs_f32_mfma_peak=round(achieved / F32_MFMA_PEAK_TFLOPS, 3),
                        traffic=traffic, algorithmic_bytes_per_launch=round(lin["bytes"] / lin["launches"]),
                        launches_per_step=lin["launches"] // args.steps,
                        avg_launch_us=round(1e3 * lin["ms"] / lin["launches"], 2),
                        gflop_per_step=round(lin["flops"] / args.steps / 1e9, 1),
                        share_of_step=round(lin["ms"] / args.steps / ms_per_step, 3),
                        measured="HIP events around each launch, over a repeat of the timed %d steps (instrumented step: %.3f ms)"
                                 % (args.steps, 1e3 * prof_elapsed / args.steps))
    cpu_baseline = None
    if n_cpu:
        from oracle import bigc_oracle as O
        cpu_sample = (synthetic_proposal(base + v, n_trajs=args.n_trajs, feat_width=width, num_enti_cats=cfg["num_enti_cats"])
                      for v in range(n_cpu))                      # the same videos, regenerated lazily on the host
        torch.set_num_threads(usable_cores())
        cpu_pairs, n_done = 0, 0
        with torch.no_grad():
            first = next(cpu_sample)
            O.forward_test(sd, cfg, [first], args.topk)                  # warm-up
            cpu_s = 0.0
            for p in [first] + list(range(n_cpu - 1)):                   # bounded: stop after ~15 s of oracle time
                p = p if not isinstance(p, int) else next(cpu_sample)
                t0 = time.perf_counter()
                O.forward_test(sd, cfg, [p], args.topk)
                cpu_s += time.perf_counter() - t0
                cpu_pairs += p.num_proposals * (p.num_proposals - 1)
                n_done += 1
                if cpu_s > 15.0:
                    break
        cpu_baseline = dict(value=round(cpu_pairs / cpu_s, 1), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                            sample="%d of the same synthetic videos (N=%d), oracle/bigc_oracle.py forward_test incl. "
                                   "construct_triplet, fp32 torch CPU, %.1f s" % (n_done, args.n_trajs, cpu_s))
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
                   "rows_per_step": packed.n_rows, "pairs_per_gpu_per_step": pairs_per_step, "parallelism": "dp%d" % world},
        "roofline": roofline, "cpu_baseline": cpu_baseline, "exact_f32_gemm_mode": exact_leg,
        "kernel_families_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in families.items()},
        "videos_per_sec": round(args.videos * world * args.steps / elapsed, 1),
        "triplets_last_step": int(n_out.sum()), "setup_s": round(gen_s, 1),
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
