"""-m gpu: what the multi-GPU path can be shown to do on ONE GPU: torch.distributed with backend "nccl" (RCCL) and
a process group of one rank, a record block pushed through the COLLECTIVE branch of distributed.gather_poses (RCCL initialisation
and a device-tensor all_gather had never executed anywhere: every multi-rank run so far was gloo and world 1 short-circuits);
and the context reporting the device it sits on (device guard)."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from accurate_aprilgroup_tracking_amd import distributed as D
r, lr, w = D.init(backend="nccl", force=True)
assert (r, w) == (0, 1) and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
assert D.backend_name().startswith("rccl")
dev = torch.device("cuda", 0)
# a record block in the real layout: [steps, streams, AGT_STATE_STRIDE] f64, written by the device tracker
from accurate_aprilgroup_tracking_amd import hiplib as H, synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
s = syn.Sequence(640, 480, n_tags=12, n_frames=5, seed=0)
frames = torch.from_numpy(s.frames()).to(dev)
trk = StreamTracker(s.width, s.height, s.obj, s.K, None, n_streams=1)
trk.pipeline(2)
trk.reset(frames[0:1].contiguous(), torch.from_numpy(s.corners(0)[None]).to(dev).contiguous())
so = trk.new_state_buffer(4)
trk.step_many(frames[1:].unsqueeze(1).contiguous(), so)
trk.join(); torch.cuda.synchronize()
assert so.cpu().numpy()[:, 0, H.ST_OK].all()
plain = D.gather_poses(so)                                  # world 1: the short cut
coll = D.gather_poses(so, force_collective=True)            # RCCL all_gather_into_tensor on the device tensor
torch.cuda.synchronize()
assert coll.is_cuda and coll.shape == (1,) + tuple(so.shape) and coll.data_ptr() != so.data_ptr()
assert torch.equal(coll, plain) and torch.equal(coll[0], so)
m = D.max_over_ranks(3.5, dev)
assert m == 3.5
D.barrier()
torch.distributed.destroy_process_group()
print("RCCL_WORLD1_OK", tuple(coll.shape))
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_rccl_world1_collective_branch_of_gather_poses():
    """in a child process (a process group is process-wide state; the suite's other tests stay undistributed)"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_context_reports_the_device_it_sits_on():
    import torch
    from accurate_aprilgroup_tracking_amd import cv_hip, hiplib as H
    ctx = cv_hip.Context(640, 480, max_level=2, max_points=48, max_streams=1)
    cus, xcds = C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(32)
    H.check(ctx.L.agt_device_info(ctx.h, C.byref(cus), C.byref(xcds), arch, 32), "agt_device_info")
    p = torch.cuda.get_device_properties(0)
    assert cus.value == p.multi_processor_count
    assert arch.value.decode() == p.gcnArchName.split(":")[0] == "gfx950"
    assert xcds.value in (1, 2, 4, 8)
    if cus.value % 32 == 0 and cus.value // 32 in (1, 2, 4, 8):
        assert xcds.value == cus.value // 32, "gfx950: one XCD per 32 CUs"
    else:
        assert xcds.value == 1


def test_two_rank_rehearsal_of_the_bench_flow_on_one_card():
    """`bench.py --gpus 2 --workload c4` exactly as the driver starts an N-GPU run without a launcher -- the parent counts
    devices without touching the GPU and starts the ranks as fresh children (bench.py self_launch: no exec of an initialised process
    anywhere) -- on the ONE card of this box: the ranks share it and gather over gloo (RCCL needs a GPU per rank), everything else is
    the N-GPU flow: rank-specific streams (seeds 1000 r + s), one gather of the record block per timed block, max-over-ranks timing,
    rank 0 printing the line.  Asserted: two ranks, the gathered block is in rank order and holds each rank's own poses, it is
    labelled a rehearsal.  No scaling figure is read off it."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AGT_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--steps", "8", "--blocks", "2",
                        "--warmup", "4", "--render-frames", "6", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "REHEARSAL" in r.stderr, "the parent says what it is about to do before it starts the ranks"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["gather_order_ok"] is True
    assert d["rehearsal"] is True and d["dist_backend"].startswith("gloo")
    assert d["gathered_shape"][0] == 2 and d["gathered_shape"][1] == 8
    assert d["config"]["workload"].startswith("c4") and d["scaling"] == "weak" and d["value"] > 0
    assert d["accepted_frac"] == 1.0 and d["gather_max_abs_pose_err_vs_truth"] < 0.05
