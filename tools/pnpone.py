#!/usr/bin/env python3
"""ONE pose solve per launch (48 corners, tracker mode with the motion-model guess: the c2 frame's pose role as a stand-alone kernel), 60
launches -- for `rocprofv3 --pmc SQ_INSTS_VALU ...` passes: dynamic instruction counts of a solve (tools/pnp_eval_isa.py has the static
per-part counts).  Prints iterations per solve so that per-evaluation figures can be derived."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
from accurate_aprilgroup_tracking_amd import synthetic as syn
from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
seq = syn.Sequence(1280, 720, n_tags=12, n_frames=12, seed=0, supersample=1)
trk = StreamTracker(1280, 720, seq.obj, seq.K, None, n_streams=1)
trk.reset()
so = trk.new_state_buffer()
its = []
for k in range(60):
    img = torch.from_numpy(seq.corners(k % 12)[None]).cuda().contiguous()
    trk.estimate_pose(img, None, so); torch.cuda.synchronize()
    its.append(int(so.cpu().numpy()[0, hiplib.ST_ITERS]))
print("iterations per solve:", its[:24], "mean of the guessed solves %.2f" % np.mean(its[2:]))
