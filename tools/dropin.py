"""drop-in latency of PoseDetector(backend='stream') on the knobs build with host-side timing of agt_track_host_frame"""
import os, sys, time, json, tempfile, logging
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accurate_aprilgroup_tracking_amd import hiplib
if os.environ.get("AGT_LIB"):
    hiplib.LIB_PATH = os.path.join(os.path.dirname(hiplib.LIB_PATH), os.environ["AGT_LIB"])
from accurate_aprilgroup_tracking_amd import formats, synthetic as syn
from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
sq = syn.Sequence(1280, 720, n_tags=12, n_frames=24, seed=0, supersample=3, group_seed=0)
tmp = tempfile.mkdtemp(); open(os.path.join(tmp, "april_group.json"), "w").write(json.dumps(sq.group))
class Det(PoseDetector):
    DIRPATH = tmp
log = logging.getLogger("b"); log.setLevel(logging.CRITICAL)
tag_ids = [int(t) for t in sq.group["tags"].keys()]
frames = sq.frames()
first = [formats.make_detection(t, c) for t, c in zip(tag_ids, sq.corners(0).reshape(-1, 4, 2))]
det = Det(log, sq.K, None, True, detector=lambda gray: first, backend="stream")
buf = det.frame_buffer(frames[0].shape)
def pp(i, nf=24):
    j = i % (2 * nf - 2); return j if j < nf else 2 * nf - 2 - j
ts = []
for k in range(310):
    np.copyto(buf, frames[pp(k)])
    t0 = time.perf_counter(); det._detect_and_get_pose(buf); dt = time.perf_counter() - t0
    if k == 0: det.detector = None
    if k >= 10: ts.append(dt)
ts = np.array(ts) * 1e6
print("drop-in gray pinned: median %.1f p10 %.1f p90 %.1f us, accepted %s" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), det.last_error is not None and det.last_error < 2))
