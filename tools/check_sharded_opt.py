"""The optimiser sharded over the ranks against the same optimiser unsharded: accept / reject sequence and loss rows.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29517 \
        tools/check_sharded_opt.py levels|stripes [--backend nccl|gloo] [--c-abi-comm|--torch-comm] [--share-gpu]

--backend nccl: one rank per GPU, RCCL (tests/test_hip_serving.py runs this when the box has two GPUs);
--backend gloo --share-gpu: rehearsal on a one-GPU box, every rank on cuda:0.
--c-abi-comm: the collectives behind the C ABI (nst_comm_*: one ncclAllReduce of the packed gradient + loss row per
closure, in stripes mode one more of the Gram / content / TV sums); the communicator id travels over the torch.distributed
process group.  World 1 runs too (a communicator of one rank; stripes: ONE stripe = the whole top level through the window
closure)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from artstyletransfer_amd.engine import Communicator, PixelOptimizer

ap = argparse.ArgumentParser()
ap.add_argument("mode", choices=["levels", "stripes"])
ap.add_argument("--backend", default="gloo")
ap.add_argument("--c-abi-comm", action="store_true")
ap.add_argument("--torch-comm", action="store_true")
ap.add_argument("--share-gpu", action="store_true")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--stripe-levels", type=int, default=2, help="stripes: how many of the top levels are cut into stripes")
args = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = 0 if args.share_gpu or args.backend == "gloo" and torch.cuda.device_count() < world else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")


NS = min(args.stripe_levels, args.levels)


def run(sharded):
    eng, x, cfg, host = bench.build_job(args.levels, 0, local)
    opt = PixelOptimizer(eng, "lbfgs", 10.0, 1)
    comm = None
    prep = lambda a: eng.prepare_img(torch.from_numpy(a).to(x.device))
    if sharded and args.c_abi_comm:
        ids = [Communicator.unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        comm = Communicator(local, rank, world, ids[0])
        if args.mode == "levels":
            opt.shard_levels_comm(comm)
        else:
            opt.shard_stripes(rank, world, host[3], [prep(host[0][l]) for l in range(NS)], [prep(host[1][l]) for l in range(NS)], comm=comm)
    elif sharded and args.mode == "levels":
        opt.shard_levels(rank, world, dist)
    elif sharded:
        opt.shard_stripes(rank, world, host[3], [prep(host[0][l]) for l in range(NS)], [prep(host[1][l]) for l in range(NS)], dist)
    totals, accepted = [], []
    for _ in range(args.steps):
        info, rows = opt.step(x, cfg.content_weight, cfg.style_weight, cfg.tv_weight)
        totals += [rows[k].copy() for k in range(len(rows))]
        accepted.append(int(info.accepted))
    out = (np.array(totals), accepted, float(x.double().sum()))
    seen = comm.info() if comm is not None else None
    opt.close()
    if comm is not None:
        comm.close()
    eng.close()
    return out, seen


(sh_rows, sh_acc, sh_sum), seen = run(world > 1 or args.c_abi_comm)
(un_rows, un_acc, un_sum), _ = run(False)
if rank == 0:
    print("world", world, args.mode, "accepted", sh_acc, "x checksum", sh_sum, "comm (rank, world, calls, bytes)", seen)
    assert sh_acc == un_acc, (sh_acc, un_acc)
    acc_rows = [i for i in range(len(sh_rows))]
    if args.mode == "levels" and world <= 2:
        # level rows have one contributor each and the total is re-formed in level order; the gradient is the sum of TWO
        # parts (commutative): bit-identical
        assert np.array_equal(sh_rows, un_rows), np.abs(sh_rows - un_rows).max()
        assert sh_sum == un_sum
    elif args.mode == "levels":
        # three and more parts are summed by the collective in another order than the unsharded chain sums them
        # (g0 + D^T(g1 + D^T(g2 + ...))): the first closures to fp32 rounding, later trial points as under stripes
        print("max rel diff of the loss rows", float(np.max(np.abs(sh_rows - un_rows) / np.maximum(np.abs(un_rows), 1e-30))),
              "checksum rel diff", abs(sh_sum - un_sum) / abs(un_sum))
        np.testing.assert_allclose(sh_rows[:3], un_rows[:3], rtol=1e-5)
        ratio = sh_rows[:, -1] / un_rows[:, -1]
        assert np.all((ratio > 0.5) & (ratio < 2.0)), ratio
        assert abs(sh_sum - un_sum) <= 1e-9 * abs(un_sum)
    else:
        # stripes: gradients near a stripe boundary are sums of two separately rounded parts
        np.testing.assert_allclose(sh_rows[:3], un_rows[:3], rtol=1e-5)
        # The later rows are rejected trial points of the shipped L-BFGS semantics: a step of lr = 10 along a
        # direction scaled by (y.s)/(y.y) with y = g1 - g0 the difference of two nearly equal gradients (the first step is
        # 1/|g|_1 long) - the cancellation amplifies the 1e-6 rounding difference of the stripe sums to percents of the
        # step length and of the (rejected) loss there.  Same decision, same order of magnitude:
        ratio = sh_rows[:, -1] / un_rows[:, -1]
        assert np.all((ratio > 0.5) & (ratio < 2.0)), ratio
        assert abs(sh_sum - un_sum) <= 1e-9 * abs(un_sum)
    print("SHARDED == UNSHARDED")
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
