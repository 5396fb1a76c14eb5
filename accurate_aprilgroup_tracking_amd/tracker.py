"""StreamTracker: B independent streams tracked entirely on the device.

One `step(frames)` = pyramid(new frames) -> LK(previous corners) -> solvePnP(guess) ->
reprojection gate -> motion-model guess update, i.e. PoseDetector._estimate_pose
(detect_pose.py:467-574) fed by LK-tracked corners, with the PoseDetector state
(extrinsic_guess, prev_transform, velocity buffers) resident in HBM.  Four kernel launches
per step, no host<->device copy and no synchronisation; read `state` back when convenient.
"""
import ctypes as C
import numpy as np
import torch

from . import hiplib as H
from .cv_hip import Context, _ptr, _host_f64


class TrackState(C.Structure):
    """Mirror of csrc/agt_kernels.h AgtTrackState (tests read it back)."""
    _fields_ = [("guess", C.c_double * 6), ("prev", C.c_double * 6), ("rot_vel", (C.c_double * 9) * 2),
                ("tran_vel", (C.c_double * 3) * 2), ("prev_R", C.c_double * 9), ("has_guess", C.c_int), ("has_prev", C.c_int),
                ("n_vel", C.c_int), ("frame", C.c_int), ("guess_t_f32", C.c_int), ("prev_t_f32", C.c_int),
                ("chain_fault", C.c_int), ("pad", C.c_int)]


class StreamTracker:
    def __init__(self, width, height, obj_points, K, dist=None, n_streams=1, max_level=2, win=21,
                 enhance_ape=True, reproject=False, min_points=8, gate_px=2.0, device=None):
        obj = np.ascontiguousarray(np.asarray(obj_points, np.float32).reshape(-1, 3))
        self.n = obj.shape[0]
        self.B = n_streams
        self.ctx = Context(width, height, max_level=max_level, win=win, max_points=max(self.n, 8),
                           max_streams=n_streams, device=device)
        self.dev = torch.device("cuda", self.ctx.device)
        self.obj = torch.from_numpy(obj).to(self.dev)
        self.K, _ = _host_f64(K)
        self.dist, self.ndist = _host_f64(dist)
        self.enhance_ape = bool(enhance_ape)
        assert H.lib().agt_tracker_state_size() == C.sizeof(TrackState)
        H.check(self.ctx.L.agt_tracker_options(self.ctx.h, int(reproject), int(min_points), float(gate_px)),
                "agt_tracker_options")
        self._alive = []            # frames aliased by pyramid level 0 of the ring entries in flight
        self._keep_frames = max((max_level + 6) + 2, 12)

    def _dist_ptr(self):
        return self.dist.ctypes.data_as(C.c_void_p) if self.ndist else None

    def reset(self, frames=None, corners=None):
        """frames: cuda u8 [B,H,W] in which `corners` (cuda f32 [B,n,2]) were seen.  With both None
        only `estimate_pose` (detector-supplied corners) can be used afterwards."""
        self.ctx.use_current_stream()
        # (frames in flight come first -- slot 0 is a ring entry of the running tracker: agt_pyramid_build and
        # agt_tracker_reset both drain the pipeline themselves; reset is also the recovery from AGT_ERR_CHAIN)
        if frames is not None:
            self.ctx.pyramid_build(0, frames)
            assert corners.dtype == torch.float32 and corners.is_contiguous() and corners.shape == (self.B, self.n, 2)
        H.check(self.ctx.L.agt_tracker_reset(self.ctx.h, 0, _ptr(corners), _ptr(self.obj), self.n, self.B,
                                             self.K.ctypes.data_as(C.c_void_p), self._dist_ptr(), self.ndist,
                                             int(self.enhance_ape)), "agt_tracker_reset")
        self._alive = [frames]

    def pipeline(self, depth):
        """0 / False: separate launches per stage, pose complete in stream order.  F >= 1 (True = 1, the default):
        fused software-pipelined step advancing every stage by F frames per launch -- one launch every F calls
        of step(); the record of frame t is written about (L + 2) * F steps later (join() flushes)."""
        depth = int(depth)
        H.check(self.ctx.L.agt_tracker_pipeline(self.ctx.h, depth), "agt_tracker_pipeline")
        # (big batches run their stage kernels on library-owned streams: a frame is dead 9 calls after it was handed over)
        self._keep_frames = max((self.ctx.max_level + 6) * max(depth, 1) + 2, 12)

    def step(self, frames, state_out=None):
        """frames: cuda u8 [B,H,W] (a reference is kept while the frame is in flight: level 0 of the
        pyramid ring aliases it).  state_out: cuda f64 [B, STATE_STRIDE] or None; with the pipeline on
        it is written a few steps later -- call join() before consuming it.  Enqueues only."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.shape[0] == self.B
        assert tuple(frames.shape[1:]) == (self.ctx.height, self.ctx.width) and frames.stride(2) == 1, "frames must be [B, H, W] with unit pixel stride"
        H.check(self.ctx.L.agt_track_frame(self.ctx.h, _ptr(frames), frames.stride(1), frames.stride(0), self.B,
                                           _ptr(state_out)), "agt_track_frame")
        self._alive.append(frames)
        if len(self._alive) > self._keep_frames:
            del self._alive[0]
        return state_out

    def tag_gate(self, corners_per_tag=4):
        """4: the pose solve uses a corner only while all four corners of its tag are usable, min_points = 8 then is the
        reference's ">= 2 tags" (detect_pose.py:494-496).  0: every usable corner counts (the default of the C ABI)."""
        H.check(self.ctx.L.agt_tracker_tag_gate(self.ctx.h, int(corners_per_tag)), "agt_tracker_tag_gate")

    def rewind(self):
        """Take the newest frame back as the tracking source: the next step() tracks from the frame before it (the reference
        keeps the older frame as "previous" when a frame yields no tag, detect_pose.py:570-574).  Joins the pipeline."""
        H.check(self.ctx.L.agt_tracker_rewind(self.ctx.h), "agt_tracker_rewind")

    def step_detected(self, frames, corners, mask=None, state_out=None):
        """A detector-fed frame (detect_pose.py:576-609 with >= 2 tags found): `frames` joins the stream (the next step()
        tracks FROM it), `corners` cuda f32 [B,n,2] / `mask` cuda u8 [B,n] become its corner set and LK status and
        _estimate_pose runs on them.  Enqueues only; the record is complete in stream order."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.shape[0] == self.B
        assert tuple(frames.shape[1:]) == (self.ctx.height, self.ctx.width) and frames.stride(2) == 1, "frames must be [B, H, W] with unit pixel stride"
        assert corners.dtype == torch.float32 and corners.is_contiguous() and corners.shape == (self.B, self.n, 2)
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (self.B, self.n)
        H.check(self.ctx.L.agt_track_frame_detected(self.ctx.h, _ptr(frames), frames.stride(1), frames.stride(0), self.B,
                                                    _ptr(corners), _ptr(mask), _ptr(state_out)), "agt_track_frame_detected")
        self._alive.append(frames)
        if len(self._alive) > self._keep_frames:
            del self._alive[0]
        return state_out

    def step_many(self, clip, state_out=None):
        """A clip of consecutive frames in one call: clip cuda u8 [K,B,H,W], state_out cuda f64 [K,B,STATE_STRIDE] or None.
        Same as K calls of step() (same launches, same records) without the per-call host cost.  Enqueues only."""
        assert clip.dtype == torch.uint8 and clip.is_cuda and clip.dim() == 4 and clip.shape[1] == self.B
        assert tuple(clip.shape[2:]) == (self.ctx.height, self.ctx.width) and clip.stride(3) == 1, "clip frames must be H x W with unit pixel stride"
        K = clip.shape[0]
        if state_out is not None:
            assert state_out.is_contiguous() and tuple(state_out.shape) == (K, self.B, H.STATE_STRIDE)
        H.check(self.ctx.L.agt_track_frames(self.ctx.h, _ptr(clip), clip.stride(2), clip.stride(1), clip.stride(0), self.B, K,
                                            _ptr(state_out)), "agt_track_frames")
        self._alive.append(clip)                  # the whole clip stays referenced while any of its frames may be in flight
        while len(self._alive) > self._keep_frames:
            del self._alive[0]
        return state_out

    def dense_model(self, model_xyz, model_t, iters=5, photo_weight=0.05, reseed=True):
        """BASELINE configs[4]: register the dense model (cuda f32 [M,3] object-frame samples, cuda f32 [M] template
        intensities; kept alive here) -- step_dense() then refines every accepted pose photometrically on the device.
        model_xyz None disables the stage."""
        if model_xyz is None:
            H.check(self.ctx.L.agt_tracker_dense(self.ctx.h, None, None, 0, 0, 0.0, 0), "agt_tracker_dense")
            self._dense = None
            return
        assert model_xyz.dtype == torch.float32 and model_xyz.is_cuda and model_xyz.is_contiguous() and model_xyz.shape[1] == 3
        assert model_t.dtype == torch.float32 and model_t.is_cuda and model_t.is_contiguous() and model_t.shape[0] == model_xyz.shape[0]
        H.check(self.ctx.L.agt_tracker_dense(self.ctx.h, _ptr(model_xyz), _ptr(model_t), model_xyz.shape[0], int(iters),
                                             float(photo_weight), int(bool(reseed))), "agt_tracker_dense")
        self._dense = (model_xyz, model_t)

    def step_dense(self, frames, state_out=None, dense_out=None):
        """One frame with the dense stage: pyramid -> LK -> solvePnP(guess) + gate + motion model -> dense refinement of the
        accepted pose (-> corner re-seed).  dense_out: cuda f64 [B, DENSE_STRIDE] (created when None).  Enqueues only."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.shape[0] == self.B
        assert tuple(frames.shape[1:]) == (self.ctx.height, self.ctx.width) and frames.stride(2) == 1, "frames must be [B, H, W] with unit pixel stride"
        if dense_out is None:
            dense_out = torch.zeros((self.B, H.DENSE_STRIDE), dtype=torch.float64, device=self.dev)
        H.check(self.ctx.L.agt_track_frame_dense(self.ctx.h, _ptr(frames), frames.stride(1), frames.stride(0), self.B,
                                                 _ptr(state_out), _ptr(dense_out)), "agt_track_frame_dense")
        self._alive.append(frames)
        if len(self._alive) > self._keep_frames:
            del self._alive[0]
        return dense_out

    def step_many_dense(self, clip, state_out=None, dense_out=None):
        """A clip of consecutive frames with the dense stage in one call: clip cuda u8 [K,B,H,W], state_out cuda f64
        [K,B,STATE_STRIDE] or None, dense_out cuda f64 [K,B,DENSE_STRIDE] (created when None).  Same records as K calls of
        step_dense(); the library builds each next frame's pyramid inside the current frame's dense launch.  Enqueues only."""
        assert clip.dtype == torch.uint8 and clip.is_cuda and clip.dim() == 4 and clip.shape[1] == self.B
        assert tuple(clip.shape[2:]) == (self.ctx.height, self.ctx.width) and clip.stride(3) == 1, "clip frames must be H x W with unit pixel stride"
        K = clip.shape[0]
        if dense_out is None:
            dense_out = torch.zeros((K, self.B, H.DENSE_STRIDE), dtype=torch.float64, device=self.dev)
        assert dense_out.is_contiguous() and tuple(dense_out.shape) == (K, self.B, H.DENSE_STRIDE)
        if state_out is not None:
            assert state_out.is_contiguous() and tuple(state_out.shape) == (K, self.B, H.STATE_STRIDE)
        H.check(self.ctx.L.agt_track_frames_dense(self.ctx.h, _ptr(clip), clip.stride(2), clip.stride(1), clip.stride(0), self.B, K,
                                                  _ptr(state_out), _ptr(dense_out)), "agt_track_frames_dense")
        self._alive.append(clip)
        while len(self._alive) > self._keep_frames:
            del self._alive[0]
        return dense_out

    def join(self):
        """Enqueue the remaining pipeline stages of every supplied frame (no host synchronisation)."""
        H.check(self.ctx.L.agt_tracker_join(self.ctx.h), "agt_tracker_join")

    def estimate_pose(self, corners, mask=None, state_out=None):
        """PoseDetector._estimate_pose on device state with supplied corners (cuda f32 [B,n,2])."""
        assert corners.dtype == torch.float32 and corners.is_contiguous() and corners.shape == (self.B, self.n, 2)
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (self.B, self.n)
        H.check(self.ctx.L.agt_estimate_pose(self.ctx.h, _ptr(corners), _ptr(mask), self.B, _ptr(state_out)),
                "agt_estimate_pose")
        return state_out

    def estimate_pose_from_detections(self, detections_per_stream, tag_ids, state_out=None):
        """Detector-fed frame (detect_pose.py:389-437 -> :467-574): `detections_per_stream` holds, per stream,
        the swatbotics-style detections of this frame; `tag_ids` is the tag order of the object points.
        Builds the dense corner table + mask (formats.detections_to_corner_table) and runs the device
        state machine."""
        from . import formats
        tabs = [formats.detections_to_corner_table(d, tag_ids) for d in detections_per_stream]
        assert len(tabs) == self.B and tabs[0][0].shape[0] == self.n
        corners = torch.from_numpy(np.stack([t[0] for t in tabs])).to(self.dev)
        mask = torch.from_numpy(np.stack([t[1] for t in tabs])).to(self.dev)
        return self.estimate_pose(corners, mask, state_out)

    def new_state_buffer(self, frames=1):
        shape = (frames, self.B, H.STATE_STRIDE) if frames > 1 else (self.B, H.STATE_STRIDE)
        return torch.zeros(shape, dtype=torch.float64, device=self.dev)

    def read_state(self):
        """-> list of TrackState (synchronises)."""
        buf = (TrackState * self.B)()
        H.check(self.ctx.L.agt_tracker_state_read(self.ctx.h, C.byref(buf), self.B), "agt_tracker_state_read")
        return list(buf)

    def corners(self):
        """current corner set and LK status as torch views are not exposed by pointer; copy out"""
        cp, sp = C.c_void_p(), C.c_void_p()
        H.check(self.ctx.L.agt_tracker_buffers(self.ctx.h, C.byref(cp), C.byref(sp)), "agt_tracker_buffers")
        return cp.value, sp.value
