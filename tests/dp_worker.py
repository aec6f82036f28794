"""Child process of tests/test_multi_rank_gpu.py: one RANK of a data-parallel HIP training step (the two ranks of the test share
the one GPU of the box and exchange gradients through the host with gloo; RCCL cannot put two ranks on one device -- with
argv[4] = "nccl" every rank takes its own GPU and the all-reduce runs on RCCL).
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; argv: golden case name, output file, batch size, [backend]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vidsgg-big_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import cases  # noqa: E402
from bigc_hip.shard import allreduce_gradients, shard_indices, zero_gradients  # noqa: E402
from test_hip_train import build_train_model  # noqa: E402


def main():
    name, out_path, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        # RCCL: one rank per device (torch.cuda.device_count() does not initialise the GPU; the group is created first)
        assert torch.cuda.device_count() >= world, "the nccl variant needs one GPU per rank"
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    cfg, variant, sd, props, gts = cases.make_train_case(name)
    props, gts = props[:batch], gts[:batch]
    model = build_train_model(cfg, variant, sd)
    if backend == "nccl":
        model = model.to("cuda:%d" % rank)
    mine = shard_indices(len(props), rank, world)            # utils/DataParallel.py:80-92
    params = list(model.parameters())
    if mine:
        loss, _ = model([props[i] for i in mine], [gts[i] for i in mine])
        loss.mean().backward()
        loss = float(loss.detach())
    else:
        zero_gradients(params)
        loss = 0.0
    allreduce_gradients(params, dist, contributing=min(len(props), world))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"loss": loss, "grads": {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None},
                    "in_place": allreduce_gradients.last_in_place}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
