"""bench_pairs.py -- BASELINE.json configs[2] exactly as SURVEY.md section 8d states it, for bench.py --workload c3pairs:
a batch of 64 INDEPENDENT COLD 1280x720 frame pairs per step, 48 corners each.

A step = for 64 pairs at once: build BOTH pyramids (agt_pyramid_build x 2: the previous and the next frame of every pair,
nothing cached from an earlier step), cv::calcOpticalFlowPyrLK on them (agt_lk_track), cv::solvePnP with the extrinsic guess
(agt_solve_pnp) -- the stateless C-ABI entry points, stage kernels in stream order.  Algorithmic bytes:
B_pair = 2 * W * H * 1.3125 + N * L * (24^2 + 32^2) + N * 21 = 2,650,608 B at 1280x720, L = 3, N = 48; 169,638,912 B per batch.
Four distinct batches (4 x 118 MB of frames > the 256 MiB Infinity Cache) rotate, so every step reads HBM-cold frames.

Batches are independent of each other, so consecutive steps are software-pipelined: NCTX contexts (each with its own two pyramid
slots and its own HIP stream) take the steps round-robin, and the chip overlaps the HBM-bound pyramid passes of one batch with the
VALU- / latency-bound LK and PnP kernels of the batches before it.  Inside a batch the four stages stay in stream order.
--pair-contexts 1 gives the strictly serial form (one stream).
"""
import ctypes as C
import json
import os
import time

import numpy as np

NBATCH = 4


def pair_bytes(W, H, npts, levels=3):
    return int(2 * W * H * 1.3125 + npts * levels * (24 * 24 + 32 * 32) + npts * 21)


def main_pairs(args, torch, D, HL, wl, rank, world, dev, rehearsal):
    out = measure_pairs(args, torch, D, HL, wl, rank, world, dev, rehearsal)
    if rank == 0:
        print(json.dumps(out), flush=True)
    D.barrier()


def measure_pairs(args, torch, D, HL, wl, rank, world, dev, rehearsal):
    """-> the JSON object of the c3pairs workload on rank 0 (None elsewhere); also the `pairs64_hbm` extra of the default run"""
    import bench as B_
    from accurate_aprilgroup_tracking_amd import cv_hip, synthetic as syn
    W, H, B = wl["W"], wl["H"], args.streams or wl["B"]
    K, Wm = args.steps, args.warmup
    t_r = time.time()
    nseq = 4
    NF = max(args.render_frames // 3, NBATCH + 1)
    seqs = [syn.Sequence(W, H, n_tags=wl["ntags"], n_frames=NF, seed=1000 * rank + s, supersample=3, group_seed=0) for s in range(nseq)]
    npts = seqs[0].obj.shape[0]
    # batch j, pair b: frames (k, k + 1) of sequence b % nseq with k = (j + b // nseq) % (NF - 1); every pair at its own address
    prev = torch.empty((NBATCH, B, H, W), dtype=torch.uint8, device=dev)
    nxt = torch.empty((NBATCH, B, H, W), dtype=torch.uint8, device=dev)
    pts = np.empty((NBATCH, B, npts, 2), np.float32)
    guess = np.empty((NBATCH, B, 6), np.float64)
    truth = np.empty((NBATCH, B, 6), np.float64)
    rend = [torch.from_numpy(sq.frames()).to(dev) for sq in seqs]
    for j in range(NBATCH):
        for b in range(B):
            sq, k = seqs[b % nseq], (j + b // nseq) % (NF - 1)
            prev[j, b] = rend[b % nseq][k]; nxt[j, b] = rend[b % nseq][k + 1]
            pts[j, b] = sq.corners(k)
            guess[j, b] = np.concatenate([sq.rvecs[k], sq.tvecs[k]])
            truth[j, b] = np.concatenate([sq.rvecs[k + 1], sq.tvecs[k + 1]])
    del rend
    render_s = time.time() - t_r
    pts_d = torch.from_numpy(pts).to(dev); guess_d = torch.from_numpy(guess).to(dev)
    obj_d = torch.from_numpy(seqs[0].obj.astype(np.float32)).to(dev)
    NCTX = max(1, int(getattr(args, "pair_contexts", 4) or 4))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NCTX - 1)]
    ctxs = []
    for s_ in streams:
        with torch.cuda.stream(s_):
            ctxs.append(cv_hip.Context(W, H, max_level=B_.LEVELS - 1, win=B_.WIN, max_points=npts, max_streams=B))     # bound to s_
    L = ctxs[0].L
    # every context shares the device with the other contexts' pyramid passes: cap the tracker's resident workgroups per CU (agt_hip.h
    # agt_lk_occupancy_cu; --lk-cu 0 = uncapped, the form of rounds 3-4)
    lk_occ = int(getattr(args, "lk_cu", 8)) if NCTX > 1 else 0
    for c_ in ctxs:
        HL.check(L.agt_lk_occupancy_cu(c_.h, lk_occ), "agt_lk_occupancy_cu")
    nxs = [torch.zeros((B, npts, 2), dtype=torch.float32, device=dev) for _ in range(NCTX)]
    sts = [torch.zeros((B, npts), dtype=torch.uint8, device=dev) for _ in range(NCTX)]
    pose = torch.zeros((K, B, 6), dtype=torch.float64, device=dev)           # one record per step (the gathered poses)
    infos = [torch.zeros((B, 4), dtype=torch.int32, device=dev) for _ in range(NCTX)]
    Kh = np.ascontiguousarray(seqs[0].K.reshape(-1)); Kp = Kh.ctypes.data_as(C.c_void_p)
    vp = lambda t: C.c_void_p(t.data_ptr())
    pitch, bstride = W, W * H

    PAIR_BUILD = not getattr(args, "no_pair_build", False)
    # experiment (--pnp-stream): the pose solves of ALL batches on one more context / stream, ordered by events -- a batch's stream goes on to
    # its next pair build right after the LK launch instead of sitting through a 64-wave solve
    PNP_SIDE = bool(getattr(args, "pnp_stream", False)) and NCTX > 1
    if PNP_SIDE:
        s_pnp = torch.cuda.Stream()
        with torch.cuda.stream(s_pnp):
            ctx_pnp = cv_hip.Context(W, H, max_level=B_.LEVELS - 1, win=B_.WIN, max_points=npts, max_streams=B)
        ev_lk = [torch.cuda.Event() for _ in range(NCTX)]
        ev_pnp = [torch.cuda.Event() for _ in range(NCTX)]
        pnp_pending = [False] * NCTX

    def step(j, pose_k, events=None, q=0):
        h, nx, st, info, sq_ = ctxs[q].h, nxs[q], sts[q], infos[q], streams[q]
        rec = (lambda i: events[i].record(sq_)) if events else (lambda i: None)
        rec(0)
        if PAIR_BUILD:
            # both pyramids of the batch's pairs in one call / one launch of the two-level pass over 2 B images (round 6: agt_pyramid_build_pair)
            HL.check(L.agt_pyramid_build_pair(h, vp(prev[j]), vp(nxt[j]), pitch, bstride, B), "agt_pyramid_build_pair")
            rec(1)
        else:
            HL.check(L.agt_pyramid_build(h, 0, vp(prev[j]), pitch, bstride, B), "agt_pyramid_build")
            rec(1)
            HL.check(L.agt_pyramid_build(h, 1, vp(nxt[j]), pitch, bstride, B), "agt_pyramid_build")
        rec(2)
        if PNP_SIDE and not events and pnp_pending[q]:
            sq_.wait_event(ev_pnp[q])                      # the corner arrays of this context are read by its previous batch's solve
        HL.check(L.agt_lk_track(h, 0, 1, vp(pts_d[j]), vp(nx), vp(st), None, npts, B, 3, 30, 0.01, 0, 1e-4), "agt_lk_track")
        rec(3)
        if PNP_SIDE and not events:
            ev_lk[q].record(sq_)
            s_pnp.wait_event(ev_lk[q])
            HL.check(L.agt_solve_pnp(ctx_pnp.h, vp(obj_d), 0, vp(nx), HL.F32, vp(st), npts, B, Kp, None, 0, vp(pose_k), 1, vp(info), None), "agt_solve_pnp")
            ev_pnp[q].record(s_pnp)
            pnp_pending[q] = True
            return
        # (pose_k holds the extrinsic guess of every pair of batch j on entry -- an input, put there before the timed region --
        # and the solved pose afterwards: agt_solve_pnp works in place, as cv2 does)
        HL.check(L.agt_solve_pnp(h, vp(obj_d), 0, vp(nx), HL.F32, vp(st), npts, B, Kp, None, 0, vp(pose_k), 1, vp(info), None), "agt_solve_pnp")
        rec(4)

    def load_guesses(first_step):
        """inputs of a block: pose[k] <- the guesses of the batch step first_step + k will solve (untimed, like the frames)"""
        idx = torch.as_tensor([(first_step + k) % NBATCH for k in range(K)], device=dev)
        pose.copy_(guess_d[idx])

    torch.cuda.synchronize()                  # the frames and tables above were written on the default stream
    load_guesses(0)
    torch.cuda.synchronize()
    for k in range(Wm):
        step(k % NBATCH, pose[k % K], q=k % NCTX)
    torch.cuda.synchronize()
    D.gather_poses(pose)
    dts = []
    for r in range(max(1, args.blocks)):
        load_guesses(r * K)
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        for k in range(K):
            step((r * K + k) % NBATCH, pose[k], q=k % NCTX)
        torch.cuda.synchronize()                                               # every stream: the poses of all K steps are complete
        gathered = D.gather_poses(pose)
        torch.cuda.synchronize(); D.barrier()
        dts.append(D.max_over_ranks(time.perf_counter() - t0, dev))
    med, p10, p90 = B_.percentiles(dts)
    pairs_s = world * B * K / med
    # correctness of what was timed: poses of the last block against the generator's truth, LK status
    ph = pose.cpu().numpy()
    err = max(np.abs(ph[k] - truth[((len(dts) - 1) * K + k) % NBATCH]).max() for k in range(K))
    tracked = float(np.mean([s_.cpu().numpy().mean() for s_ in sts]))
    if rank == 0:
        # instrumented pass: HIP events on the launch stream around every call (the context launches on torch's current stream)
        M = min(K, 100)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(M)]
        load_guesses(0)
        torch.cuda.synchronize()
        for k in range(M):
            step(k % NBATCH, pose[k], ev[k], q=0)                                 # serial pass on one stream: the kernels' own durations
        torch.cuda.synchronize()
        sp = np.array([[e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(4)] for e in ev]).mean(axis=0)      # us
        # ... and the same calls in the form that is TIMED: round-robin over the NCTX contexts, events on each context's own
        # stream.  An event pair around a call spans from the end of the stream's previous work to the end of the call's kernel -- the
        # kernel's duration while it shares the chip with the other contexts' kernels, plus its launch gap.
        sp_pipe = None
        if NCTX > 1:
            Mp = (min(K, 128) // NCTX) * NCTX
            evp = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(Mp)]
            load_guesses(0)
            torch.cuda.synchronize()
            for k in range(Mp):
                step(k % NBATCH, pose[k], evp[k], q=k % NCTX)
            torch.cuda.synchronize()
            sp_pipe = np.array([[e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(4)] for e in evp[NCTX:]]).mean(axis=0)      # (the first round fills the pipeline)
        ab = B_.algorithmic_bytes(W, H, npts)
        spans = {"pyramid_prev(1 launch)": float(sp[0]), "pyramid_next(1 launch)": float(sp[1]), "lk": float(sp[2]), "pnp": float(sp[3])}
        # (pyramid bytes: SURVEY 8d's W * H * 1.3125 per frame -- level 0 read once, levels 1 and 2 written once; rounds 3-4: the two single-level
        # passes read level 1 a second time, which is their own traffic, not algorithmic bytes: VERDICT r3 weak #6)
        per = {"pyramid": (2 * B * ab["pyramid"], float(sp[0] + sp[1])), "lk": (B * ab["lk"], float(sp[2])), "pnp": (B * ab["pnp"], float(sp[3]))}
        # The DOMINANT kernel of this HBM workload = the one that moves the step's bytes: the pyramid pass, 91 % of the 169.6 MB (and the
        # longest stage of the serial pass until round 5 capped the tracker's occupancy, which stretches the LK launch ALONE from 36 to
        # 45 us while the pipelined step gets shorter: the longest launch of the serial pass is no longer the kernel that bounds the step)
        dom = max(per, key=lambda n: per[n][0])
        nlaunches = {"pyramid": 1 if PAIR_BUILD else 2, "lk": 1, "pnp": 1}              # (round 5: one two-level pass per pyramid build; round 6: one per PAIR of builds)
        nlaunch = nlaunches[dom]
        kernel_us_alone = per[dom][1] / nlaunch
        achieved_alone = per[dom][0] / nlaunch / (kernel_us_alone * 1e-6) / 1e9
        # the line's roofline figure is the kernel AS TIMED (pipelined pass); the stand-alone figure rides beside it
        pipe_us = {"pyramid": float(sp_pipe[0] + sp_pipe[1]) / nlaunches["pyramid"], "lk": float(sp_pipe[2]), "pnp": float(sp_pipe[3])} if sp_pipe is not None else None
        kernel_us = pipe_us[dom] if pipe_us else kernel_us_alone
        achieved = per[dom][0] / nlaunch / (kernel_us * 1e-6) / 1e9
        every = {n: {"avg_launch_us": round(per[n][1] / nlaunches[n], 3), "bytes_per_launch": int(per[n][0] / nlaunches[n]),
                     "algorithmic_GBs": round(per[n][0] / nlaunches[n] / (per[n][1] / nlaunches[n] * 1e-6) / 1e9, 1)} for n in per}
        batch_bytes = B * pair_bytes(W, H, npts)
        whole = batch_bytes / (med / K) / 1e9
        roof = {"bound": "hbm", "kernel": {"pyramid": ("pyr_group_kernel (the two-level register-rolling pass, alternating strip directions: L0->L1->L2 of both frames of 64 pairs = 128 frames in ONE launch per step)" if PAIR_BUILD else "pyr_roll2_kernel (two pyrDown levels per pass, register-rolling, alternating strip directions: L0->L1->L2 of 64 frames per launch, 2 launches per step)"), "lk": "lk_kernel<21,1,3> (one wave per corner)",
                                           "pnp": "pnp_kernel<float,1>"}[dom],
                "achieved": round(achieved, 3), "peak": B_.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / B_.HBM_PEAK_GBS, 6),
                "traffic": B_.pmc_traffic({"pyramid": "pyr_group_kernel c3pairs (pair build)" if PAIR_BUILD else "pyr_roll2_kernel c3pairs", "lk": "lk_kernel<21,1,3> c3pairs", "pnp": "pnp_kernel c3pairs"}[dom], kernel_us_alone),
                "avg_launch_us": round(kernel_us, 3), "bytes_per_launch": int(per[dom][0] / nlaunch),
                "measured": ("pipelined pass (the form that is timed: %d contexts round-robin, HIP events on each context's stream around the call)" % NCTX) if pipe_us else "serial pass (one context)",
                "alone": {"avg_launch_us": round(kernel_us_alone, 3), "achieved": round(achieved_alone, 3), "frac": round(achieved_alone / B_.HBM_PEAK_GBS, 6),
                          "measured": "serial pass on one stream: the kernel with the chip to itself"},
                "call_spans_us_pipelined_pass": ({k_: round(v, 2) for k_, v in zip(("pyramid_prev(1 launch)", "pyramid_next(1 launch)", "lk", "pnp"), sp_pipe.tolist())} if sp_pipe is not None else None),
                "whole_step": {"algorithmic_GBs": round(whole, 1), "frac_of_8TBs": round(whole / B_.HBM_PEAK_GBS, 4), "bytes_per_step": int(batch_bytes),
                               "frac_of_measured_copy_6290GBs": round(whole / 6290.0, 4)},
                "call_spans_us_serial_pass": {k_: round(v, 2) for k_, v in spans.items()}, "every_kernel_serial_pass": every}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_pairs(seqs, NF)
        out = {"metric": "frame pairs/sec (both pyramids + LK + PnP) on cold 1280x720 pairs", "value": round(pairs_s, 2), "unit": "pairs/s",
               "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(med / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8/i64 (LK), f64 (PnP)", "data": "synthetic",
               "config": {"workload": wl["label"] % B, "pairs_per_step": B, "frames_resident": "%d batches x %d pairs x 2 frames in HBM (%.0f MiB), rotated"
                          % (NBATCH, B, NBATCH * B * 2 * W * H / 2**20),
                          "launch": "stateless C-ABI calls, per batch in stream order: %s, agt_lk_track, " % ("agt_pyramid_build_pair (one two-level pass over both frames of every pair)" if PAIR_BUILD else "agt_pyramid_build x 2 (one two-level pass each)") +
                                    "agt_solve_pnp (guess); consecutive batches round-robin over %d contexts / HIP streams (software pipelining across independent batches)" % NCTX,
                          "contexts": NCTX, "lk_workgroups_per_cu_cap": lk_occ},
               "timing": {"blocks": len(dts), "steps_per_block": K, "statistic": "median block, max over ranks per block",
                          "ms_per_step_p10": round(p10 / K * 1e3, 5), "ms_per_step_p90": round(p90 / K * 1e3, 5)},
               "roofline": roof, "cpu_baseline": cpu, "max_abs_pose_err_vs_truth": float(err), "tracked_frac": round(tracked, 4),
               "render_s": round(render_s, 1), "gathered_shape": list(gathered.shape), "rccl_ranks": world, "dist_backend": D.backend_name()}
        if rehearsal:
            out["rehearsal"] = True
        return out
    return None


def cpu_baseline_pairs(seqs, NF):
    """the same pairs on ONE host core with the oracle ("port"), ~10 s: calcOpticalFlowPyrLK (both pyramids, Scharr image
    per level, per-point LK: OpenCV's dataflow) + solvePnP with the guess; then with 16 threads (bands / points)"""
    import bench as B_
    cvo, flags = B_.native_oracle()
    obj32 = seqs[0].obj.astype(np.float32)

    def run(seconds, nthreads):
        cvo.lib().cvo_set_num_threads(nthreads)
        n = 0; t0 = time.perf_counter()
        try:
            while True:
                sq = seqs[n % len(seqs)]; k = (n // len(seqs)) % (NF - 1)
                nx, st, _ = cvo.calcOpticalFlowPyrLK(sq.frame(k), sq.frame(k + 1), sq.corners(k), maxLevel=2, nthreads=nthreads)
                m = st.ravel() == 1
                cvo.solvePnP(obj32[m], nx.reshape(-1, 2)[m], sq.K, None, sq.rvecs[k].copy(), sq.tvecs[k].copy(), True)
                n += 1
                if time.perf_counter() - t0 > seconds:
                    break
        finally:
            cvo.lib().cvo_set_num_threads(1)
        return n, time.perf_counter() - t0
    n1, dt1 = run(10.0, 1)
    curve = [{"cores": 1, "value": round(n1 / dt1, 2)}]
    for nthr in B_.cpu_thread_counts()[1:]:            # 1 thread and every available core, with the steps between
        n2, dt2 = run(2.5, nthr)
        curve.append({"cores": nthr, "value": round(n2 / dt2, 2), "sample": "%d pairs, %.1f s" % (n2, dt2)})
    top = curve[-1]
    return {"value": round(n1 / dt1, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d cold pairs of the same sequences: oracle calcOpticalFlowPyrLK (two pyramids + Scharr + LK) + solvePnP(guess), 1 thread, %.1f s; host has %d cores, %d available to this process"
                      % (n1, dt1, os.cpu_count(), B_.cpu_available()),
            "cpu_model": B_.cpu_model(), "build": flags,
            "all_cores": {"value": top["value"], "cores": top["cores"], "sample": top.get("sample", "")} if len(curve) > 1 else None,
            "threads_curve": curve, "best_of_curve": max(curve, key=lambda e_: e_["value"]), "cgroup_cpu_quota_cores": B_.cpu_quota()}
