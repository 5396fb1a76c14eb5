#!/usr/bin/env python3
"""bench.py -- frames/sec of the LK + iterative-PnP hot path on MI355X.

Workload (BASELINE.json configs[1], "c2"): ONE 1280x720 synthetic dodeca stream per GPU,
12 tags / 48 corners, 3-level LK pyramid (maxLevel=2), 21x21 window, COUNT+EPS (30, 0.01),
iterative PnP with the motion-model extrinsic guess.  A step = one frame of the stream:
pyramid(new frame) -> LK(prev corners) -> solvePnP(guess) -> gate -> motion model, all on
the device (agt_track_frame), frames already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3]
    N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract: task prompt, section 4).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REDETECT = 600            # frames between detector refreshes of the corner set (drift control, see time_tracker)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
W, H, NTAGS, NPTS, LEVELS, WIN = 1280, 720, 12, 48, 3, 21


def algorithmic_bytes():
    """SURVEY.md 8d per-unit figures (1280x720, L=3, N=48), per launch of each kernel."""
    pyr = sum((W >> l) * (H >> l) + (W >> (l + 1)) * (H >> (l + 1)) for l in range(LEVELS - 1))   # read l, write l+1
    lk = NPTS * LEVELS * (24 * 24 + 32 * 32) + NPTS * (8 + 8 + 1 + 4)
    pnp = NPTS * 20 + 48
    return {"pyramid": pyr, "lk": lk, "pnp": pnp, "frame": W * H * 1.3125 + NPTS * LEVELS * 1600 + NPTS * 21}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json;
    FETCH_SIZE / WRITE_SIZE are collected in their own runs, never inside this timed program)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[kernel]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pingpong(i, nf):
    j = i % (2 * nf - 2)
    return j if j < nf else 2 * nf - 2 - j


def build_stream_ring(torch, syn, dev, rank, B, NF):
    """Render NF frames for up to 4 seeds and lay a ping-pong sequence over an HBM ring that exceeds the
    256 MiB Infinity Cache, so every step reads cold addresses."""
    nseq = min(B, 4)
    seqs = [syn.Sequence(W, H, n_tags=NTAGS, n_frames=NF, seed=1000 * rank + s, supersample=3, group_seed=0) for s in range(nseq)]
    rendered = np.stack([sq.frames() for sq in seqs], axis=1)          # [NF, nseq, H, W]
    period = 2 * NF - 2
    ring_slots = period * max(1, -(-(300 << 20) // (period * B * W * H)))
    ring = torch.empty((ring_slots, B, H, W), dtype=torch.uint8, device=dev)
    src = torch.from_numpy(rendered).to(dev)
    for i in range(ring_slots):
        f = src[pingpong(i, NF)]
        for b in range(B):
            ring[i, b] = f[b % nseq]       # streams beyond the rendered seeds are copies at distinct HBM addresses
    corners0 = np.stack([seqs[b % nseq].corners(0) for b in range(B)])
    # ground-truth corners of every frame of the ping-pong period: what a detector pass on that frame would return
    truth = np.stack([np.stack([seqs[b % nseq].corners(pingpong(i, NF)) for b in range(B)]) for i in range(period)])
    build_stream_ring.truth = torch.from_numpy(truth).to(dev).contiguous()
    return seqs, rendered, ring, ring_slots, corners0


def time_tracker(torch, D, HL, trk, ring, ring_slots, corners0, dev, B, K, Wm, world):
    """W warm-up + K timed steps between barrier + synchronize pairs; returns (fps, dt, state[K+W,B,16])."""
    state = torch.zeros((Wm + K, B, HL.STATE_STRIDE), dtype=torch.float64, device=dev)

    truth = build_stream_ring.truth
    since = [0]

    def run(n_steps, first, st):
        # Raw LK chaining drifts (the reference re-detects the tags on every frame): after REDETECT frames the corners
        # are refreshed from the detector's answer for the current frame (ground truth here), as a hybrid pipeline would.
        for k in range(n_steps):
            if since[0] >= REDETECT:
                j = (first + k - 1) % ring_slots
                trk.join()
                trk.reset(ring[j], truth[j % truth.shape[0]])
                since[0] = 0
            trk.step(ring[(first + k) % ring_slots], st[k] if st is not None else None)
            since[0] += 1
    trk.reset(ring[0], torch.from_numpy(corners0).to(dev).contiguous())
    run(Wm, 1, state[:Wm])
    trk.join()
    D.gather_poses(state[:Wm])                    # warm the communicator outside the timed region
    torch.cuda.synchronize(); D.barrier()
    t0 = time.perf_counter()
    run(K, 1 + Wm, state[Wm:])
    trk.join()                                    # enqueue the last pipeline stages of the frames in flight
    # the only collective: the per-frame state records (128 B per stream-frame), once per chunk of
    # K frames.  state[Wm:] is already contiguous: no torch kernel runs inside the timed region.
    gathered = D.gather_poses(state[Wm:])
    torch.cuda.synchronize(); D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    return world * B * K / dt, dt, state, gathered, run


def event_spans(trk, HL, run, ring, corners0, torch, dev, Wm, M, pipeline):
    """average HIP-event spans (us) of M steps, recorded on the launch stream by the library"""
    trk.pipeline(pipeline)
    trk.reset(ring[0], torch.from_numpy(corners0).to(dev).contiguous())
    run(Wm, 1, None)
    trk.join()
    HL.check(trk.ctx.L.agt_profile_begin(trk.ctx.h, M), "agt_profile_begin")
    run(M, 1 + Wm, None)
    ms = np.zeros((M, HL.PROF_SPANS), np.float32); nrec = C.c_int(0)
    HL.check(trk.ctx.L.agt_profile_end(trk.ctx.h, ms.ctypes.data_as(C.c_void_p), C.byref(nrec)), "agt_profile_end")
    trk.join()
    return ms[:nrec.value].mean(axis=0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"])
    ap.add_argument("--streams", type=int, default=None, help="independent streams per GPU (c2: 1, c3: 64)")
    ap.add_argument("--render-frames", type=int, default=24)
    ap.add_argument("--depth", type=int, default=0,
                    help="frames per fused launch (agt_tracker_pipeline depth; 1 = lowest latency); 0 = 4 for >= 100 timed steps, "
                         "2 for >= 40, else 1 (short runs are dominated by filling and draining the pipeline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-extra", action="store_true", help="skip the 64-stream HBM-bound side measurement")
    args = ap.parse_args()

    import torch
    from accurate_aprilgroup_tracking_amd import distributed as D
    rank, local_rank, world = D.init()
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1)     # > 1 rank per GPU only in gloo rehearsals (AGT_DIST_BACKEND)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from accurate_aprilgroup_tracking_amd import hiplib as HL, synthetic as syn
    from accurate_aprilgroup_tracking_amd.tracker import StreamTracker

    B = args.streams or (1 if args.workload == "c2" else 64)
    K, Wm, NF = args.steps, args.warmup, args.render_frames

    t_r = time.time()
    seqs, rendered, ring, ring_slots, corners0 = build_stream_ring(torch, syn, dev, rank, B, NF)
    render_s = time.time() - t_r
    sq0 = seqs[0]
    trk = StreamTracker(W, H, sq0.obj, sq0.K, None, n_streams=B, max_level=LEVELS - 1, win=WIN, enhance_ape=True)
    fused = B * NPTS <= 2048           # fused launch (agt_step_fits)
    auto_depth = 4 if args.steps >= 100 else (2 if args.steps >= 40 else 1)
    depth = max(1, min(args.depth or auto_depth, 8)) if fused else 1
    trk.pipeline(depth)
    fps, dt, state, gathered, run = time_tracker(torch, D, HL, trk, ring, ring_slots, corners0, dev, B, K, Wm, world)
    st = state.cpu().numpy()
    accepted = float(st[Wm:, :, HL.ST_OK].mean())
    iters = float(st[Wm:, :, HL.ST_ITERS].mean())

    if rank == 0:
        ab = algorithmic_bytes()
        M = max(depth, min(K, 200) // depth * depth)      # instrumented passes: whole launch groups, at least one
        # per-launch durations from HIP events on the launch stream, second (instrumented) pass
        stage_us = event_spans(trk, HL, run, ring, corners0, torch, dev, Wm, M, False)     # separate kernels
        names = ["pyramid", "lk", "pnp"]
        if fused:
            # one launch per `depth` steps: average launch period from two HIP events around M / depth launches on
            # the launch stream (steady state: every launch carries depth frames of each pipeline stage)
            def fused_period(d):
                trk.pipeline(d)
                trk.reset(ring[0], torch.from_numpy(corners0).to(dev).contiguous())
                run(Wm // d * d, 1, None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(M // d * d, 1 + Wm // d * d, None); e1.record()
                trk.join(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / (M // d * d) * d
            launch_us = fused_period(depth)
            launch_us_d1 = fused_period(1) if depth != 1 else launch_us
            achieved = depth * B * ab["frame"] / (launch_us * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": "step_kernel<21,4,3> (fused: PnP | LK | pyrDown x2, %d consecutive frames of each stage per launch)" % depth,
                    "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": pmc_traffic("step_kernel<21,4,3> depth %d" % depth) if B == 1 else None,
                    "frac_of_measured_copy_6290GBs": round(achieved / 6290.0, 6),
                    "avg_launch_us": round(launch_us, 3), "frames_per_launch": depth, "bytes_per_launch": int(depth * B * ab["frame"]),
                    "one_frame_per_launch": {"avg_launch_us": round(launch_us_d1, 3), "frames_per_s": round(B / (launch_us_d1 * 1e-6), 1)},
                    "note": "latency-bound by construction: one 720p stream is a serial chain of ~10 LK and 4 LM iterations"}
        else:
            dom = int(np.argmax(stage_us))
            launches = {"pyramid": LEVELS - 1, "lk": 1, "pnp": 1}[names[dom]]
            kernel_us = float(stage_us[dom]) / launches
            achieved = B * ab[names[dom]] / launches / (kernel_us * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": {"pyramid": "pyr_down_kernel", "lk": "lk_kernel<21,1,3>", "pnp": "pnp_kernel<float,1>"}[names[dom]],
                    "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": None, "avg_launch_us": round(kernel_us, 3), "bytes_per_launch": int(B * ab[names[dom]] / launches)}
        roof["separate_kernel_spans_us"] = {n: round(float(v), 3) for n, v in zip(names, stage_us)}

        extra = None
        if args.workload == "c2" and world == 1 and not args.no_batch_extra:
            extra = batch_extra(torch, D, HL, syn, StreamTracker, dev, rank, NF)

        cpu = pose_err = None
        if not args.no_cpu_baseline and world == 1:
            cpu, pose_err = cpu_baseline(seqs[0], rendered[:, 0], st[:, 0], Wm, K, NF)
        out = {"metric": "frames/sec (LK+PnP) on 1280x720 dodeca stream", "value": round(fps, 2), "unit": "frames/s",
               "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(dt / K * 1e3, 5),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i64 (LK), f64 (PnP)",
               "data": "synthetic",
               "config": {"workload": "%s: %d x 1280x720 stream(s) per GPU, 12 tags/48 corners, 3-level LK 21x21, "
                                      "iterative PnP with motion-model guess" % (args.workload, B),
                          "streams_per_gpu": B, "frames_resident": "HBM ring %d slots (%.0f MiB)" % (ring_slots, ring.numel() / 2**20),
                          "parallelism": "stream-per-GPU x%d, RCCL all_gather of poses once" % world,
                          "launch": ("fused software-pipelined step, %d frames per launch (record of frame t written ~%d steps later)"
                                     % (depth, (LEVELS + 1) * depth)) if fused else "separate kernels per stage"},
               "roofline": roof, "cpu_baseline": cpu,
               "pose_err_vs_cpu": pose_err, "accepted_frac": round(accepted, 4), "mean_lm_iters": round(iters, 2),
               "batch64_hbm": extra, "render_s": round(render_s, 1), "gathered_shape": list(gathered.shape)}
        print(json.dumps(out), flush=True)
    D.barrier()


def batch_extra(torch, D, HL, syn, StreamTracker, dev, rank, NF):
    """BASELINE.json configs[2]-style side measurement: 64 independent 1280x720 streams per step (stage kernels on three
    overlapped library streams; the spans come from a second, serial pass).  Reports whole-step frames/s and the pyrDown
    kernel's HBM rate."""
    B, K, Wm = 64, 60, 10
    seqs, rendered, ring, ring_slots, corners0 = build_stream_ring(torch, syn, dev, rank, B, min(NF, 8))
    trk = StreamTracker(W, H, seqs[0].obj, seqs[0].K, None, n_streams=B, max_level=LEVELS - 1, win=WIN, enhance_ape=True)
    fps, dt, state, _, run = time_tracker(torch, D, HL, trk, ring, ring_slots, corners0, dev, B, K, Wm, 1)
    spans = event_spans(trk, HL, run, ring, corners0, torch, dev, Wm, 40, False)
    ab = algorithmic_bytes()
    pyr_gbs = B * ab["pyramid"] / (float(spans[0]) * 1e-6) / 1e9
    ok = float(state.cpu().numpy()[Wm:, :, HL.ST_OK].mean())
    del ring
    torch.cuda.empty_cache()
    return {"workload": "64 x 1280x720 streams per step, stage kernels overlapped on three streams (span_us: serial pass)", "frames_per_s": round(fps, 1),
            "ms_per_step": round(dt / K * 1e3, 4), "span_us": {"pyramid(2 launches)": round(float(spans[0]), 2),
                                                                "lk": round(float(spans[1]), 2), "pnp": round(float(spans[2]), 2)},
            "pyr_down_algorithmic_GBs": round(pyr_gbs, 1), "pyr_down_frac_of_8TBs": round(pyr_gbs / HBM_PEAK_GBS, 4),
            "accepted_frac": round(ok, 4)}


def cpu_baseline(seq, frames, gpu_state, Wm, K, NF):
    """Reference CPU path stand-in ("port"): the oracle's cvo_track_frame (full padded pyramids,
    full-frame Scharr image per level, per-point LK, FP64 LM) on ONE host core, on a bounded
    sample of the same stream; plus the pose difference HIP vs CPU chain on identical frames."""
    import logging, tempfile
    from oracle import cvoracle as cvo, cv2_shim
    from accurate_aprilgroup_tracking_amd import hiplib as HL
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    cvo.build()
    # -- timing: ~10 s of CPU work
    pyr = cvo.Pyramid(frames[0]); pts = seq.corners(0)
    r, t = seq.rvecs[0].copy(), seq.tvecs[0].copy()
    n = 0; t0 = time.perf_counter()
    while True:
        k = pingpong(n + 1, NF)
        pyr, pts, stt, er, cnt, r, t = cvo.track_frame(pyr, frames[k], pts, seq.obj, seq.K, None, r, t, nthreads=1)
        n += 1
        if time.perf_counter() - t0 > 10.0 or n >= 20000:
            break
    dt = time.perf_counter() - t0
    # same chain with OpenMP over rows / points on every host core (as OpenCV's parallel_for_), ~5 s
    try:
        nthr = len(os.sched_getaffinity(0))
    except AttributeError:
        nthr = os.cpu_count() or 1
    nthr = max(1, min(nthr, 16))          # the GPU box grants about 16 cores per GPU; 48 corners do not scale further anyway
    pyr2 = cvo.Pyramid(frames[0]); pts2 = seq.corners(0)
    r2, t2 = seq.rvecs[0].copy(), seq.tvecs[0].copy()
    n2 = 0; t1 = time.perf_counter()
    while nthr > 1:
        k = pingpong(n2 + 1, NF)
        pyr2, pts2, _, _, _, r2, t2 = cvo.track_frame(pyr2, frames[k], pts2, seq.obj, seq.K, None, r2, t2, nthreads=nthr)
        n2 += 1
        if time.perf_counter() - t1 > 5.0 or n2 >= 4000:
            break
    dt2 = time.perf_counter() - t1
    cpu = {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same 1280x720 stream, oracle cvo_track_frame (pyramid+Scharr+LK+LM), 1 thread, %.1f s; host has %d cores"
                     % (n, dt, os.cpu_count()),
           "cpu_model": cpu_model(),
           "all_cores": {"value": round(n2 / dt2, 2), "cores": nthr, "sample": "%d frames, %.1f s" % (n2, dt2)} if n2 else None}
    # -- parity: reference-validated state machine on the oracle backend over the first frames
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "april_group.json"), "w").write(json.dumps(seq.group))

    class Det(PoseDetector):
        DIRPATH = tmp
    log = logging.getLogger("bench"); log.setLevel(logging.CRITICAL)
    det = Det(log, seq.K, None, True, cv=cv2_shim.make_cv2())
    obj32 = seq.obj.astype(np.float32)
    pyr = cvo.Pyramid(frames[0]); pts = seq.corners(0)
    nchk = min(Wm + K, 60)
    dr = dtv = 0.0
    for i in range(nchk):
        npyr = cvo.Pyramid(frames[pingpong(i + 1, NF)])
        nx, status, _ = cvo.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2)
        nx = nx.reshape(-1, 2); status = status.ravel()
        il = [nx[j].reshape(1, 1, 2) for j in range(len(nx)) if status[j]]
        ol = [obj32[j].reshape(1, 3) for j in range(len(nx)) if status[j]]
        det._estimate_pose(il if len(il) >= 8 else [], ol if len(il) >= 8 else [])
        if det.last_pose[0] is not None and gpu_state[i, HL.ST_OK]:
            dr = max(dr, float(np.linalg.norm(gpu_state[i, :3] - det.last_pose[0].ravel())))
            dtv = max(dtv, float(np.linalg.norm(gpu_state[i, 3:6] - det.last_pose[1].ravel().astype(np.float64))))
        pts = nx.astype(np.float32); pyr = npyr
    return cpu, {"max_l2_drvec": dr, "max_l2_dtvec": dtv, "frames": nchk, "tolerance": 1e-4}


if __name__ == "__main__":
    main()
