"""Which keyframes a step trains on: KeyframeScheduler (host/include/keyframe_scheduler.h) = the bookkeeping of
GaussianMapper::useOneRandomSlidingWindowKeyframe (src/gaussian_mapper.cpp:1126-1173) on keyframe indices, and its batch form
for the data-parallel step ("the reference scheduler called B times on rank 0 and broadcast as indices")."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    return load_host("emu")


def reference_walk(order, remaining, used, cursor):
    """useOneRandomSlidingWindowKeyframe (:1139-1163) restated on plain lists: the next keyframe of the shuffled cycle that has
    uses left; a full cycle without one grants every keyframe one more use."""
    start = cursor
    while True:
        cursor += 1
        if cursor >= len(order):
            cursor = 0
        if cursor == start:
            for k in range(len(remaining)):
                remaining[k] += 1
        kf = order[cursor]
        if remaining[kf] > 0:
            break
    used[kf] += 1
    remaining[kf] -= 1
    return kf, cursor


def test_scheduler_walks_like_the_reference():
    ops = _ops()
    h = ops.keyframe_scheduler_create(11)
    assert ops.keyframe_scheduler_use_one(h) == -1                    # no keyframes: nullptr (:1129-1130)
    n, uses = 7, [3, 1, 4, 1, 5, 2, 2]
    for u in uses:
        ops.keyframe_scheduler_add(h, u)
    draws = [int(ops.keyframe_scheduler_use_one(h)) for _ in range(60)]
    # every keyframe's uses are spent before the refill, in the order of ONE permutation walked cyclically
    first_cycle = draws[:n]
    assert sorted(first_cycle) == list(range(n)), first_cycle          # the first n draws visit every keyframe once: a permutation
    # (the cursor is pre-incremented: the first draw is the permutation's element 1, the n-th its element 0)
    order = [first_cycle[-1]] + first_cycle[:-1]
    remaining, used, cursor = list(uses), [0] * n, 0
    want = []
    for _ in range(60):
        kf, cursor = reference_walk(order, remaining, used, cursor)
        want.append(kf)
    assert draws == want
    state = [int(x) for x in ops.keyframe_scheduler_state(h)]
    assert state[:n] == used and state[n:] == remaining and sum(used) == 60
    # more uses for one keyframe (local BA / loop closure: :841, :930): it is drawn until they are spent like the others
    ops.keyframe_scheduler_increase(h, 3, 9)
    remaining[3] += 9
    for _ in range(25):
        kf, cursor = reference_walk(order, remaining, used, cursor)
        assert int(ops.keyframe_scheduler_use_one(h)) == kf
    # a new keyframe: the permutation is made again (addKeyframe clears kfid_shuffled_), every index stays reachable
    ops.keyframe_scheduler_add(h, 2)
    seen = {int(ops.keyframe_scheduler_use_one(h)) for _ in range(200)}
    assert seen == set(range(n + 1))
    # two schedulers with one seed and the same calls agree; another seed gives another permutation
    a, b, c = (ops.keyframe_scheduler_create(s) for s in (5, 5, 6))
    for hh in (a, b, c):
        for _ in range(12):
            ops.keyframe_scheduler_add(hh, 2)
    da, db, dc = ([int(x) for x in ops.keyframe_scheduler_use_batch(hh, 12)] for hh in (a, b, c))
    assert da == db and sorted(da) == list(range(12)) and da != dc


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
entry.load_package()
sys.path.insert(0, os.path.join(sys.argv[1], "photo-slam_amd", "host"))
import build_host
torch.ops.load_library(build_host.build("emu"))
ops = torch.ops.photoslam_amd
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
h = ops.keyframe_scheduler_create(int(sys.argv[3]) + (rank if sys.argv[4] == "diverged" else 0))
for k in range(20):
    ops.keyframe_scheduler_add(h, 1 + k % 3)
batches = []
try:
    for step in range(1 if sys.argv[4] == "diverged" else 6):   # (after a divergence the ranks no longer issue the same collectives)
        batches.append([int(x) for x in ops.keyframe_scheduler_use_batch_on_ranks(h, dist.group.WORLD.group_name)])
        if step == 2:
            ops.keyframe_scheduler_add(h, 4)     # a keyframe arrives in the middle of the session (on every rank)
    np.savez(os.path.join(sys.argv[2], f"sched{rank}.npz"), batches=np.array(batches), state=np.array(ops.keyframe_scheduler_state(h)))
except RuntimeError as e:
    open(os.path.join(sys.argv[2], f"error{rank}.txt"), "w").write(str(e))
dist.barrier()
'''


@pytest.mark.parametrize("ranks", [2, 8])
def test_a_batch_of_keyframes_is_drawn_once_for_all_ranks_gloo(tmp_path, ranks):
    """B ranks, ONE seeded scheduler replicated on each: rank 0's draw is broadcast, every rank books the same B keyframes (its own
    is batch[rank]), the bookkeeping stays identical on all of them -- over six steps with a keyframe added in between; the batch
    equals B consecutive draws of a single-process scheduler."""
    _ops()   # (the emulator host library is built before the ranks start)
    script = tmp_path / "worker_sched.py"
    script.write_text(WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                           "--master-port", str(29570 + ranks), str(script), ROOT, str(tmp_path), "17", "same"],
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    outs = [np.load(tmp_path / f"sched{r}.npz") for r in range(ranks)]
    for o in outs[1:]:
        assert np.array_equal(o["batches"], outs[0]["batches"]) and np.array_equal(o["state"], outs[0]["state"])
    ops = torch.ops.photoslam_amd
    h = ops.keyframe_scheduler_create(17)
    for k in range(20):
        ops.keyframe_scheduler_add(h, 1 + k % 3)
    for step in range(6):
        assert [int(x) for x in ops.keyframe_scheduler_use_batch(h, ranks)] == list(outs[0]["batches"][step])
        if step == 2:
            ops.keyframe_scheduler_add(h, 4)
    b0 = outs[0]["batches"][0]
    assert len(set(b0.tolist())) == ranks      # (20 keyframes with uses left: a batch holds B different ones)


def test_diverged_replicas_are_reported_gloo(tmp_path):
    """Ranks whose schedulers were seeded differently: the ranks that drew another batch than rank 0 raise instead of training the
    wrong keyframe."""
    _ops()
    script = tmp_path / "worker_sched.py"
    script.write_text(WORKER)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29579", str(script), ROOT, str(tmp_path), "17", "diverged"],
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert "diverged" in (tmp_path / "error1.txt").read_text() and not (tmp_path / "error0.txt").exists()
