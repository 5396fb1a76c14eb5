"""CPU: the N > 1 path (stream sharding + the single pose gather) with world_size 2 on gloo."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from accurate_aprilgroup_tracking_amd import distributed as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _pose_block(streams, frames):
    """deterministic stand-in for tracker output: [frames, streams_local, 8]"""
    out = np.zeros((frames, len(streams), 8))
    for j, sidx in enumerate(streams):
        for f in range(frames):
            out[f, j] = [sidx + 0.1, sidx + 0.2, sidx + 0.3, f * 0.01, f * 0.02, 0.3 + sidx, 1.0, f]
    return out


def _worker(rank, world, port, n_streams, frames, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = D.shard_streams(n_streams, r, w)
    local = torch.from_numpy(_pose_block(list(mine), frames))
    D.barrier()
    allp = D.gather_poses(local)
    assert allp.shape == (world, frames, len(mine), 8)
    for q in range(world):
        ref = _pose_block(list(D.shard_streams(n_streams, q, world)), frames)
        assert np.array_equal(allp[q].numpy(), ref)
    m = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert m == float(world)
    ret[rank] = True
    torch.distributed.destroy_process_group()


def test_shard_streams_partition():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            parts = [list(D.shard_streams(n, r, w)) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_gather_poses_world2_gloo():
    world, port = 2, _free_port()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 8, 5, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_single_process_is_passthrough():
    t = torch.arange(24, dtype=torch.float64).reshape(3, 1, 8)
    assert torch.equal(D.gather_poses(t), t.unsqueeze(0))
    assert D.max_over_ranks(2.5, torch.device("cpu")) == 2.5
