#!/usr/bin/env python3
"""bench.py -- frames/sec of the LK + iterative-PnP hot path on MI355X.

Default workload (BASELINE.json configs[1], "c2"): ONE 1280x720 synthetic dodeca stream per GPU,
12 tags / 48 corners, 3-level LK pyramid (maxLevel=2), 21x21 window, COUNT+EPS (30, 0.01),
iterative PnP with the motion-model extrinsic guess.  A step = one frame of every stream of the rank:
pyramid(new frame) -> LK(prev corners) -> solvePnP(guess) -> gate -> motion model, all on
the device (agt_track_frame), frames already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c3|c3pairs|c4|c5]

    --gpus N > 1 invoked plainly: this process starts N fresh ranks (python -m torch.distributed.run, rendezvous on
    127.0.0.1) BEFORE anything touches a GPU, relays rank 0's JSON line and exits with the job's code.  Launched by
    torch.distributed.run itself (WORLD_SIZE set) it is one of the ranks.

Workloads: c2 = 1 x 1280x720 stream (configs[1], the metric's configuration); c3 = 64 x 1280x720 STREAMS per step (configs[2]
read as streams: previous pyramid cached, 1,441,008 B per stream-frame); c3pairs = configs[2] exactly as SURVEY.md 8d states it:
64 COLD frame pairs per step, both pyramids built, 2,650,608 B per pair = 169,638,912 B per batch (bench_pairs.py); c4 = 1 x 1920x1080 stream per GPU (configs[3]); c5 = 1280x720, 60 tags / 240 corners + dense photometric
refinement per frame (configs[4]).

Timing: W untimed warm-up steps, then R blocks (default 15) of EXACTLY K steps each, every block bracketed by barrier +
torch.cuda.synchronize() on both sides and including the pipeline drain and the one pose gather; the per-block time is the MAX
over ranks.  ms_per_step / value come from the MEDIAN block; p10 / p90 are reported beside it.

Prints ONE JSON line on rank 0 (contract: task prompt, section 4).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REDETECT = 600            # frames between detector refreshes of the corner set (drift control, see Bench.refresh)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
LEVELS, WIN = 3, 21

WORKLOADS = {
    #        frame size    tags  streams/GPU
    "c2": dict(W=1280, H=720, ntags=12, B=1, dense=False,
               label="c2: %d x 1280x720 stream(s) per GPU, 12 tags/48 corners, 3-level LK 21x21, iterative PnP with motion-model guess"),
    "c3": dict(W=1280, H=720, ntags=12, B=64, dense=False,
               label="c3: %d x 1280x720 independent streams per step, 48 corners each (LK HBM-bandwidth run)"),
    "c3pairs": dict(W=1280, H=720, ntags=12, B=64, dense=False, pairs=True,
                    label="c3pairs: %d independent cold 1280x720 frame pairs per step (both pyramids built, LK, PnP with guess), 48 corners each"),
    "c4": dict(W=1920, H=1080, ntags=12, B=1, dense=False,
               label="c4: %d x 1920x1080 stream(s) per GPU (one per GPU across the node), 48 corners, 3-level LK 21x21, iterative PnP"),
    "c5": dict(W=1280, H=720, ntags=60, B=1, dense=True,
               label="c5: %d x 1280x720 stream(s), 60 tags/240 corners, LK + iterative PnP + dense photometric refinement (61,440 samples)"),
}


def algorithmic_bytes(W, H, npts):
    """SURVEY.md 8d per-unit figures (L = 3), per launch of each kernel and per streamed frame.
    "pyramid" is 8d's figure: level 0 read once, every coarser level written once = W * H * 1.3125 (the
    line used to count the two single-level passes' own traffic, level 1 read a second time, which is "pyramid_two_pass")."""
    pyr = sum((W >> l) * (H >> l) for l in range(LEVELS))
    two_pass = sum((W >> l) * (H >> l) + (W >> (l + 1)) * (H >> (l + 1)) for l in range(LEVELS - 1))   # read l, write l + 1, per pass
    lk = npts * LEVELS * (24 * 24 + 32 * 32) + npts * (8 + 8 + 1 + 4)
    pnp = npts * 20 + 48
    return {"pyramid": pyr, "pyramid_two_pass": two_pass, "lk": lk, "pnp": pnp, "frame": W * H * 1.3125 + npts * LEVELS * 1600 + npts * 21}


def pmc_traffic(kernel, launch_us=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json;
    FETCH_SIZE / WRITE_SIZE are collected in their own runs, never inside this timed program).  The entry records the launch
    duration the kernel had when the counters were taken: a figure whose kernel has since changed by more than 15 % in duration
    is withheld (null) rather than reported stale."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f)[kernel]
        ref = e.get("launch_us_at_collection")
        if launch_us is not None and ref and abs(launch_us / ref - 1.0) > 0.15:
            return None
        return e["traffic_bytes_per_launch"]
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


def percentiles(samples):
    a = np.sort(np.asarray(samples, np.float64))
    return float(np.median(a)), float(np.percentile(a, 10)), float(np.percentile(a, 90))


# ---------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes.  Nothing in this (parent)
    process has touched a GPU: torch is imported only to COUNT devices (does not initialise the runtime)."""
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        env["AGT_DIST_BACKEND"] = "gloo"
    elif ndev < args.gpus and "AGT_DIST_BACKEND" not in env:
        if args.gpus > 6:
            print("bench.py: --gpus %d on a host with %d GPU(s): more ranks than may share one card" % (args.gpus, ndev), file=sys.stderr)
            sys.exit(2)
        # rehearsal of the multi-rank flow: ranks share the card(s), gather over gloo (RCCL needs one GPU per rank)
        env["AGT_DIST_BACKEND"] = "gloo"
        print("bench.py: %d GPU(s) for %d ranks -> REHEARSAL (ranks share GPUs, gloo gather); not a scaling measurement"
              % (ndev, args.gpus), file=sys.stderr)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def stream_seed(rank, s):
    """seed of rendered stream s of rank `rank` (rank r renders seeds 1000 r + s)"""
    return 1000 * rank + s


_traj = {}


def truth_pose(seed, NF, z0, frame):
    """generator's pose of stream `seed` at `frame` (trajectory only: nothing is rendered) -> [6]"""
    from accurate_aprilgroup_tracking_amd import synthetic as syn
    if (seed, NF, z0) not in _traj:
        rv, tv = syn.trajectory(NF, seed, t0=(0.01, -0.02, z0))
        _traj[(seed, NF, z0)] = np.concatenate([rv, tv], axis=1)
    return _traj[(seed, NF, z0)][frame]


def gather_order_check(gathered, world, B, nseq, NF, z0, ring_start, ring_slots, tol):
    """Multi-GPU readiness: block q of the gathered records must carry RANK q's streams, in the real record
    layout: stream b of rank q shows seed 1000 q + b % nseq, so its pose at step k must be that trajectory's pose at the
    frame of step k (ring index ring_start + 1 + k).  -> (ok, worst gap over the accepted records)"""
    g = np.asarray(gathered)
    worst, ok = 0.0, g.shape[0] == world
    for q in range(g.shape[0]):
        for b in range(B):
            for k in range(g.shape[1]):
                if g[q, k, b, 6] != 1.0:
                    continue
                fr = pingpong((ring_start + 1 + k) % ring_slots, NF)
                worst = max(worst, float(np.abs(g[q, k, b, :6] - truth_pose(stream_seed(q, b % nseq), NF, z0, fr)).max()))
    return bool(ok and worst < tol), worst


class StubTracker:
    """--dry-run only (tests/test_distributed.py): stands in for StreamTracker on a host without a GPU so that the launcher,
    the rendezvous, the block timing and the pose gather of THIS file run under gloo.  Writes records in the REAL layout
    (AGT_STATE_STRIDE doubles: the generator's pose of this rank's stream at the frame handed over, ok = 1, frame counter in
    AGT_ST_NTRACK's place), so the gather-order check of the real run is exercised on the CPU."""

    def __init__(self, rank, NF, nseq):
        self.rank, self.t, self.NF, self.nseq = rank, 0, NF, nseq

    def pipeline(self, depth):
        pass

    def reset(self, frames=None, corners=None):
        self.t = 0

    def step(self, frames, out=None):
        self.t += 1
        if out is not None:
            out.zero_()
            ring_index = int(frames[0, 0, 0])               # (the dry-run ring's pixels hold their ring index)
            for b in range(out.shape[0]):
                out[b, :6] = self.torch_from(truth_pose(stream_seed(self.rank, b % self.nseq), self.NF, 0.30, pingpong(ring_index, self.NF)))
            out[:, 6] = 1.0; out[:, 8] = self.t

    @staticmethod
    def torch_from(a):
        import torch
        return torch.from_numpy(np.asarray(a, np.float64))

    def join(self):
        pass


class Bench:
    """One rank's streams: rendered frames laid over an HBM ring, a StreamTracker, and the timed blocks."""

    def __init__(self, torch, wl, args, rank, world, dev):
        from accurate_aprilgroup_tracking_amd import synthetic as syn
        from accurate_aprilgroup_tracking_amd.tracker import StreamTracker
        self.torch, self.dev, self.rank, self.world = torch, dev, rank, world
        self.W, self.H, self.B = wl["W"], wl["H"], args.streams or wl["B"]
        self.K, self.Wm, self.NF = args.steps, args.warmup, args.render_frames
        self.sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
        W, H, B, NF = self.W, self.H, self.B, self.NF
        t_r = time.time()
        self.nseq = min(B, 4)
        self.z0 = 0.30
        if args.dry_run:
            self.NF = 3
            self.period = self.ring_slots = 4
            self.ring = torch.zeros((4, B, 8, 8), dtype=torch.uint8)
            for i in range(4):
                self.ring[i] = i
            self.truth = torch.zeros((4, B, 48, 2))
            self.npts, self.render_s, self.seqs, self.rendered = 48, 0.0, [], None
            self.trk = StubTracker(rank, self.NF, self.nseq)
            self._views = {}
            self.pos = self.since = 0
            self.clips = False
            return
        # up to 4 rendered seeds; further streams are copies at distinct HBM addresses.  The ring exceeds the 256 MiB
        # Infinity Cache, so every step reads cold addresses.
        nseq = self.nseq
        if wl["ntags"] > 12:
            self.z0 = syn.DENSE_Z0
        self.seqs = [syn.Sequence(W, H, n_tags=wl["ntags"], n_frames=NF, seed=stream_seed(rank, s), supersample=3, group_seed=0)
                     for s in range(nseq)]
        self.rendered = np.stack([sq.frames() for sq in self.seqs], axis=1)          # [NF, nseq, H, W]
        self.period = 2 * NF - 2
        self.ring_slots = self.period * max(1, -(-(300 << 20) // (self.period * B * W * H)))
        self.ring = torch.empty((self.ring_slots, B, H, W), dtype=torch.uint8, device=dev)
        src = torch.from_numpy(self.rendered).to(dev)
        for i in range(self.ring_slots):
            f = src[pingpong(i, NF)]
            for b in range(B):
                self.ring[i, b] = f[b % nseq]
        del src
        self.npts = self.seqs[0].obj.shape[0]
        # ground-truth corners of every frame of the ping-pong period: what a detector pass on that frame would return
        truth = np.stack([np.stack([self.seqs[b % nseq].corners(pingpong(i, NF)) for b in range(B)]) for i in range(self.period)])
        self.truth = torch.from_numpy(truth).to(dev).contiguous()
        self.render_s = time.time() - t_r
        sq0 = self.seqs[0]
        self.trk = StreamTracker(W, H, sq0.obj, sq0.K, None, n_streams=B, max_level=LEVELS - 1, win=WIN, enhance_ape=True)
        if getattr(args, "stream_lk_cu", -1) != -1:
            from accurate_aprilgroup_tracking_amd import hiplib as _HL
            _HL.check(self.trk.ctx.L.agt_lk_occupancy_cu(self.trk.ctx.h, int(args.stream_lk_cu)), "agt_lk_occupancy_cu")
        self._views = {}
        self.pos = 0            # ring index of the newest frame handed to the tracker
        self.clips = not args.per_step_calls
        self.since = 0          # frames since the corner set was last refreshed

    # -- stream control
    def restart(self):
        self.trk.reset(self.ring[0], self.truth[0])
        self.pos, self.since = 0, 0

    def refresh(self):
        """Raw LK chaining drifts (the reference re-detects the tags on every frame): the corners are refreshed from the
        detector's answer for the current frame (ground truth here), as a hybrid pipeline would."""
        self.trk.join()
        self.trk.reset(self.ring[self.pos % self.ring_slots], self.truth[self.pos % self.period])
        self.since = 0

    def run(self, n, out):
        """n steps; out: [n, B, 16] state records or None.  A detector refresh falls inside only when n > REDETECT.
        Consecutive ring entries go to the tracker as one clip (agt_track_frames: the same n steps, same launches and records
        as n calls of step(), without ~5 us of Python per call -- a third of the device time of a 720p frame)."""
        k = 0
        while k < n:
            if self.since >= 2 * REDETECT:
                self.refresh()
            a = (self.pos + 1) % self.ring_slots
            m = min(n - k, self.ring_slots - a, 2 * REDETECT - self.since)
            if m > 1 and self.clips:
                # (the views of the ring / of the record buffer are made once: slicing two tensors costs this harness ~4 us per call,
                # with the GPU idle -- a 20-step block is one call)
                ck = (a, m)
                clip = self._views.get(ck)
                if clip is None:
                    clip = self._views[ck] = self.ring[a:a + m]
                ov = None
                if out is not None:
                    ok_ = (id(out), k, m)
                    ov = self._views.get(ok_)
                    if ov is None:
                        ov = self._views[ok_] = out[k:k + m]
                        self._views[("keep", id(out))] = out          # (the id stays unique while the tensor is alive)
                self.trk.step_many(clip, ov)
            else:
                m = 1
                self.trk.step(self.ring[a], out[k] if out is not None else None)
            self.pos += m; self.since += m; k += m

    # -- the measurement the contract asks for
    def timed_blocks(self, D, R):
        torch, K = self.torch, self.K
        state_w = torch.zeros((max(self.Wm, 1), self.B, 16), dtype=torch.float64, device=self.dev)
        state = torch.zeros((K, self.B, 16), dtype=torch.float64, device=self.dev)
        self.restart()
        self.run(self.Wm, state_w[:self.Wm])
        self.trk.join()
        D.gather_poses(state_w)                       # warm the communicator outside the timed region
        first = None
        dts = []
        self.chain_timeouts = 0
        for r in range(R):
            if r > 0 and self.since + K > REDETECT:
                self.refresh()                        # detector work: between blocks, outside the timers
            self.sync(); D.barrier()
            self.block_start = self.pos % self.ring_slots         # ring index of the frame before the block's first step
            t0 = time.perf_counter()
            self.run(K, state)
            self.trk.join()                           # enqueue the last pipeline stages of the frames in flight
            # the only collective: the per-frame state records (128 B per stream-frame), once per block of K frames
            gathered = D.gather_poses(state)
            self.sync(); D.barrier()
            dts.append(D.max_over_ranks(time.perf_counter() - t0, self.dev))
            blk = state.cpu().numpy()
            if blk.shape[2] > 11:
                self.chain_timeouts += int(((blk[:, :, 11].astype(np.int64) & 512) != 0).sum())       # AGT_ST_FLAGS & AGT_TRK_CHAIN_TIMEOUT, EVERY block
            if first is None:
                first = blk.copy()
        st_last = state.cpu().numpy()
        return dts, state_w.cpu().numpy(), first, st_last, gathered

    def launch_period_us(self, depth, M):
        """average launch period of the fused step from two HIP events on the launch stream around M steps (steady
        state: every launch carries `depth` frames of each pipeline stage)"""
        torch = self.torch
        self.trk.pipeline(depth)
        self.restart()
        n_w = max(self.Wm // depth, 1) * depth
        self.run(n_w, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(M // depth, 1) * depth
        e0.record(); self.run(n, None); e1.record()
        self.trk.join(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n * depth

    def stage_spans_us(self, HL, M):
        """average HIP-event spans (us) of the separate stage kernels over M steps, recorded on the launch stream"""
        self.trk.pipeline(0)
        self.restart()
        self.run(self.Wm, None)
        self.trk.join()
        HL.check(self.trk.ctx.L.agt_profile_begin(self.trk.ctx.h, M), "agt_profile_begin")
        self.run(M, None)
        ms = np.zeros((M, HL.PROF_SPANS), np.float32); nrec = C.c_int(0)
        HL.check(self.trk.ctx.L.agt_profile_end(self.trk.ctx.h, ms.ctypes.data_as(C.c_void_p), C.byref(nrec)), "agt_profile_end")
        self.trk.join()
        return ms[:nrec.value, :3].mean(axis=0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--stream-lk-cu", type=int, default=-1, help="stream workloads (c2..c5): agt_lk_occupancy_cu of the tracker's context (experiment; -1 = the library's choice, 0 = none)")
    ap.add_argument("--pnp-stream", action="store_true", help="c3pairs experiment: the pose solves of all batches on one more context / stream")
    ap.add_argument("--lk-cu", type=int, default=8, help="c3pairs: cap of the LK kernel's resident workgroups per CU in every context (agt_lk_occupancy_cu; 0 = none)")
    ap.add_argument("--no-pair-build", action="store_true", help="c3pairs: two agt_pyramid_build calls per batch (the form of rounds 3-5) instead of one agt_pyramid_build_pair")
    ap.add_argument("--pair-contexts", type=int, default=4, help="c3pairs: contexts / streams the independent batches are pipelined over (1 = serial)")
    ap.add_argument("--per-step-calls", action="store_true", help="hand the frames over one agt_track_frame call at a time instead of as clips")
    ap.add_argument("--blocks", type=int, default=15, help="timed blocks of --steps steps each (median / p10 / p90 over them)")
    ap.add_argument("--streams", type=int, default=None, help="independent streams per GPU (default: the workload's)")
    ap.add_argument("--render-frames", type=int, default=24)
    ap.add_argument("--depth", type=int, default=0,
                    help="frames per fused launch (agt_tracker_pipeline depth; 1 = lowest latency); 0 = min(steps, 32): the launches are "
                         "chained (PnP follows LK inside the launch), so a short block is one launch deep (measured at 20 steps: "
                         "53.7 k / 55.6 k / 56.5 k frames/s at depth 10 / 16 / 20)")
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST HOOK (CPU, gloo): launcher + rendezvous + block timing + gather with a stub tracker; the line it prints "
                         "is marked data = dry-run and is not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-batch-extra", dest="no_extras", action="store_true",
                    help="skip the side measurements (64-stream HBM-bound step, H2D-inclusive rate, per-call latencies)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                         # does not return

    # The cold-pair side measurement is a child process with four HIP streams of its own; it runs FIRST, before this process has touched the
    # GPU: beside an idle parent that holds contexts and queues the same command read 2 % slower on one box (43.18 against 42.34 us per step,
    # profiles/r06_final_c2k20_bench.json / r06_final_c3pairs_bench.json) -- the process then has the device to itself, as the stand-alone workload does
    pre_pairs = None
    if (args.workload == "c2" and not args.no_extras and not args.dry_run and args.gpus == 1
            and int(os.environ.get("WORLD_SIZE", "1")) == 1):
        pre_pairs = pairs_extra(args)

    import torch
    from accurate_aprilgroup_tracking_amd import distributed as D
    rank, local_rank, world = D.init()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, torch, D, rank, world)
    dev_index = local_rank % max(torch.cuda.device_count(), 1)     # > 1 rank per GPU only in gloo rehearsals (AGT_DIST_BACKEND)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rehearsal = world > 1 and os.environ.get("AGT_DIST_BACKEND") == "gloo"

    from accurate_aprilgroup_tracking_amd import hiplib as HL
    wl = WORKLOADS[args.workload]
    if wl["dense"]:
        from bench_c5 import main_c5                     # configs[4]: its step has a fourth stage (bench_c5.py)
        return main_c5(args, torch, D, HL, wl, rank, world, dev, rehearsal)
    if wl.get("pairs"):
        from bench_pairs import main_pairs               # configs[2] as cold pairs (bench_pairs.py)
        return main_pairs(args, torch, D, HL, wl, rank, world, dev, rehearsal)
    bench = Bench(torch, wl, args, rank, world, dev)
    B, K, Wm, NPTS = bench.B, bench.K, bench.Wm, bench.npts
    fused = B * NPTS <= 256            # fused launch (agt_step_fits)
    auto_depth = min(K, 32)           # chained launches: a short block is ONE launch deep (plus the pyramid launch ahead of it)
    depth = max(1, min(args.depth or (auto_depth if fused else min(K, 16)), 32))      # split mode (not fused): two launches per group of `depth` frames
    bench.trk.pipeline(depth)
    dts, st_warm, st_first, st_last, gathered = bench.timed_blocks(D, max(1, args.blocks))
    med, p10, p90 = percentiles(dts)
    fps = world * B * K / med
    accepted = float(st_last[:, :, HL.ST_OK].mean())
    iters = float(st_last[:, :, HL.ST_ITERS].mean())

    if rank == 0:
        ab = algorithmic_bytes(bench.W, bench.H, NPTS)
        M = max(depth, min(K, 200) // depth * depth)      # instrumented passes: whole launch groups, at least one
        stage_us = bench.stage_spans_us(HL, M)            # separate kernels, second (instrumented) pass
        names = ["pyramid", "lk", "pnp"]
        if fused:
            launch_us = bench.launch_period_us(depth, M)
            launch_us_d1 = bench.launch_period_us(1, M) if depth != 1 else launch_us
            achieved = depth * B * ab["frame"] / (launch_us * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": "step_kernel<21,4,3> (fused: LK | PnP chained to it | pyrDown of the next group, %d consecutive frames per launch)" % depth,
                    "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": pmc_traffic("step_kernel<21,4,3> depth %d" % depth, launch_us) if (B == 1 and args.workload == "c2") else None,
                    "frac_of_measured_copy_6290GBs": round(achieved / 6290.0, 6),
                    "avg_launch_us": round(launch_us, 3), "frames_per_launch": depth, "bytes_per_launch": int(depth * B * ab["frame"]),
                    "one_frame_per_launch": {"avg_launch_us": round(launch_us_d1, 3), "frames_per_s": round(B / (launch_us_d1 * 1e-6), 1)},
                    "note": "latency-bound by construction: one stream is a serial chain of ~10 LK and 4 LM iterations per frame"}
        else:
            dom = int(np.argmax(stage_us))
            launches = {"pyramid": LEVELS - 1, "lk": 1, "pnp": 1}[names[dom]]
            kernel_us = float(stage_us[dom]) / launches
            achieved = B * ab[names[dom]] / launches / (kernel_us * 1e-6) / 1e9
            whole = B * ab["frame"] / (med / K) / 1e9
            roof = {"bound": "hbm", "kernel": {"pyramid": "pyr_group_kernel (register-rolling pyrDown, both levels in one pass since round 5)", "lk": "lk_kernel<21,1,3>", "pnp": "pnp_kernel<float,1>"}[names[dom]],
                    "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": None, "avg_launch_us": round(kernel_us, 3), "bytes_per_launch": int(B * ab[names[dom]] / launches),
                    "whole_step": {"algorithmic_GBs": round(whole, 1), "frac_of_8TBs": round(whole / HBM_PEAK_GBS, 4),
                                   "bytes_per_step": int(B * ab["frame"])}}
        roof["separate_kernel_spans_us"] = {n: round(float(v), 3) for n, v in zip(names, stage_us)}

        extras = {}
        if args.workload == "c2" and world == 1 and not args.no_extras:
            # (order matters: the 64-stream step runs its stages on three library streams beside the caller's; a fifth stream --
            # the copy stream of the H2D measurement -- would make two of them share a hardware queue and serialise)
            ring = bench.ring; bench.ring = None
            extras["batch64_hbm"] = batch_extra(torch, D, HL, args, rank, dev)
            extras["pairs64_hbm"] = pre_pairs if pre_pairs is not None else pairs_extra(args)
            extras["c5_dense240"] = c5_extra(args)
            bench.ring = ring
            bench.trk.pipeline(depth)
            extras["h2d_inclusive"] = h2d_inclusive(torch, bench, K)
            extras["per_call_latency_us"] = per_call_latency(bench)
            extras["live_frame_latency_us"] = live_latency(bench)
            extras["drop_in_frame_latency_us"] = drop_in_latency(bench)

        cpu = pose_err = None
        if not args.no_cpu_baseline and world == 1:
            cpu, pose_err = cpu_baseline(bench.seqs[0], bench.rendered[:, 0], np.concatenate([st_warm[:Wm, 0], st_first[:, 0]]), Wm, K, bench.NF)
        out = {"metric": "frames/sec (LK+PnP) on 1280x720 dodeca stream", "value": round(fps, 2), "unit": "frames/s",
               "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(med / K * 1e3, 5),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i64 (LK), f64 (PnP)",
               "data": "synthetic",
               "config": {"workload": wl["label"] % B, "streams_per_gpu": B,
                          "frames_resident": "HBM ring %d slots (%.0f MiB)" % (bench.ring_slots, bench.ring_slots * B * bench.W * bench.H / 2**20),
                          "parallelism": "stream-per-GPU x%d, %s all_gather of poses once per block" % (world, "gloo (REHEARSAL: ranks share GPUs)" if rehearsal else "RCCL"),
                          "launch": ("fused chained step, %d frames per launch: pyramid of the next group | LK | PnP one frame behind LK (arrival counters); "
                                     "frames handed over as clips (agt_track_frames)%s" % (depth, " -- one call per frame" if args.per_step_calls else "")) if fused else
                                    "split pipeline: pyramid launch per group (caller's stream) | per-frame LK launches, half the streams each on two library streams | PnP launch per group (library stream), %d frames per group" % depth},
               "timing": {"blocks": len(dts), "steps_per_block": K, "statistic": "median block, max over ranks per block",
                          "ms_per_step_p10": round(p10 / K * 1e3, 5), "ms_per_step_p90": round(p90 / K * 1e3, 5),
                          "value_p10": round(world * B * K / p90, 2), "value_p90": round(world * B * K / p10, 2)},
               "roofline": roof, "cpu_baseline": cpu,
               "pose_err_vs_cpu": pose_err, "accepted_frac": round(accepted, 4), "mean_lm_iters": round(iters, 2),
               "chain_timeouts": int(bench.chain_timeouts),
               "render_s": round(bench.render_s, 1), "gathered_shape": list(gathered.shape)}
        out.update(extras)
        # `value` is a clip rate (frames handed over in clips, `depth` per launch); the figures comparable to the
        # reference's live frame-by-frame loop sit beside it at the top level
        out["pose_latency_frames_max"] = 2 * depth if fused else 4 * depth      # pyramid launch -> chained LK | PnP launch (split: pyramid, LK, PnP a group apart + slack)
        if fused:
            out["one_frame_per_launch_fps"] = roof["one_frame_per_launch"]["frames_per_s"]
        if "live_frame_latency_us" in extras:
            out["live_frame_us_median"] = extras["live_frame_latency_us"]["median"]
        if "drop_in_frame_latency_us" in extras:
            out["drop_in_frame_us_median"] = extras["drop_in_frame_latency_us"]["gray_pinned"]["median"]
        if cpu:
            best = cpu.get("best_of_curve") or {"cores": 1, "value": cpu["value"]}
            out["vs_cpu_baseline"] = {"x_1_core": round(fps / cpu["value"], 1),
                                      "x_best_of_thread_curve": {"threads": best["cores"], "x": round(fps / best["value"], 1)},
                                      "note": "CPU port at 1 thread and at the best point of its thread curve (1 / 4 / 16 / 64 / every core listed: cpu_baseline.threads_curve; "
                                              "the boxes grant a job ~16 cores' worth of time, past that the OpenMP teams oversubscribe and collapse); "
                                              "a reported ratio, not the target (the roofline fraction is)"}
        out["rccl_ranks"] = world
        out["dist_backend"] = D.backend_name()
        if world > 1:
            # block q of the gathered records carries rank q's streams (seeds 1000 q + s): poses against each stream's own trajectory
            ok_g, worst = gather_order_check(gathered.cpu().numpy(), world, B, bench.nseq, bench.NF, bench.z0, bench.block_start, bench.ring_slots, 5e-3)
            out["gather_order_ok"] = ok_g
            out["gather_max_abs_pose_err_vs_truth"] = worst
        if rehearsal:
            out["rehearsal"] = True
        print(json.dumps(out), flush=True)
    D.barrier()


def dry_run(args, torch, D, rank, world):
    """--dry-run: the multi-rank flow of this file without a GPU (see StubTracker)."""
    bench = Bench(torch, WORKLOADS[args.workload], args, rank, world, torch.device("cpu"))
    dts, _, first, last, gathered = bench.timed_blocks(D, max(1, args.blocks))
    med, p10, p90 = percentiles(dts)
    g = gathered.cpu().numpy()
    # every rank's block came back from the gather, in rank order (block q = the streams of rank q: seeds 1000 q + s), with the
    # frame numbers of the LAST block -- the same check the real multi-rank run applies to its records
    ok, _ = gather_order_check(g, world, bench.B, bench.nseq, bench.NF, bench.z0, bench.block_start, bench.ring_slots, 1e-12)
    ok = ok and bool((np.diff(g[0, :, 0, 8]) == 1).all()) and all(np.abs(g[q, :, :, :6] - g[0, :, :, :6]).max() > 1e-4 for q in range(1, world))
    if rank == 0:
        print(json.dumps({"metric": "frames/sec (LK+PnP) on 1280x720 dodeca stream", "value": round(world * bench.B * bench.K / med, 2),
                          "unit": "frames/s", "n_gpus": world, "steps": bench.K, "warmup": bench.Wm, "ms_per_step": round(med / bench.K * 1e3, 5),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run (stub tracker, no GPU work: NOT a measurement)",
                          "config": {"workload": "dry-run", "geometry_of": args.workload}, "timing": {"blocks": len(dts), "steps_per_block": bench.K},
                          "gathered_shape": list(gathered.shape), "gather_ok": bool(ok), "dry_run": True}), flush=True)
    D.barrier()
    if not ok:
        raise SystemExit(3)


# ---------------------------------------------------------------------------------------------------------------
def h2d_inclusive(torch, bench, K):
    """the rate INCLUDING the upload of every frame from pinned host memory (the reference's frames originate
    on the host, detect_pose.py:669-681).  A copy stream uploads chunk c+1 (G frames, one hipMemcpyAsync) while the tracker
    works on chunk c; events order upload -> step and step -> buffer reuse.  Never `value`."""
    W, H, B = bench.W, bench.H, bench.B
    G, NB, depth = 8, 8, 8                          # frames per upload, chunk buffers on the device, frames per fused launch
    lag = -(-((LEVELS + 2) * depth) // G) + 1       # chunks after which a chunk's frames are dead (pipeline depth, rounded up)
    P = int(np.lcm(bench.period, G))                # whole ping-pong periods AND whole chunks: the sequence stays continuous across the wrap
    assert P + 1 <= bench.ring_slots
    host = torch.empty((P, B, H, W), dtype=torch.uint8).pin_memory()
    host.copy_(bench.ring[1:P + 1].cpu())           # stream order: frame k of the run is ring[(k + 1) % slots]
    devb = torch.empty((NB, G, B, H, W), dtype=torch.uint8, device=bench.dev)
    copy_s = torch.cuda.Stream()
    main_s = torch.cuda.Stream()                    # the tracker runs on a stream of its own here (not the legacy default stream)
    up = [torch.cuda.Event() for _ in range(NB)]
    done = [torch.cuda.Event() for _ in range(NB)]
    n = max(4, 1000 // P) * P
    state = torch.zeros((n, B, 16), dtype=torch.float64, device=bench.dev)
    torch.cuda.synchronize()
    ctx_stream = torch.cuda.stream(main_s)
    ctx_stream.__enter__()
    bench.trk.pipeline(depth)
    bench.restart()

    # uploads go through the library's own staging call (agt_upload = hipMemcpyAsync on the stream of a second, tiny context
    # bound to the copy stream): the torch route (stream context manager + tensor slicing + copy_) costs ~250 us of host time
    # per chunk and made this loop host-bound at 18 k frames/s
    from accurate_aprilgroup_tracking_amd import cv_hip, hiplib as HL
    with torch.cuda.stream(copy_s):
        up_ctx = cv_hip.Context(64, 64, max_level=0)
    chunk_bytes = G * B * H * W
    hbase, dbase = host.data_ptr(), devb.data_ptr()

    tsteps = [0.0]

    def go(chunks, out, c0):
        for c in range(c0, c0 + chunks):
            cb = c % NB
            # a frame is dead (LEVELS + 2) * depth steps after its own (the LK of the next frame reads it last): chunk
            # c - NB is dead once chunk c - NB + lag has run; its `done` event was recorded NB - lag chunks ago
            # (waited for on the HOST: a cross-stream wait in front of hipMemcpyAsync makes the runtime block the calling thread
            # until the event has fired, 300 us per chunk here; with NB - lag chunks of slack the event is long complete)
            if c - c0 >= NB:
                done[(c - NB + lag) % NB].synchronize()
            h0 = (c * G) % P
            HL.check(up_ctx.L.agt_upload(up_ctx.h, C.c_void_p(dbase + cb * chunk_bytes), C.c_void_p(hbase + h0 * B * H * W), chunk_bytes), "agt_upload")
            up[cb].record(copy_s)
            main_s.wait_event(up[cb])
            tq = time.perf_counter()
            k0 = (c - c0) * G                              # the chunk as one clip (agt_track_frames): one launch per chunk
            bench.trk.step_many(devb[cb], out[k0:k0 + G] if out is not None else None)
            bench.since += G
            tsteps[0] += time.perf_counter() - tq
            done[cb].record(main_s)
    go(20, None, 0)
    bench.trk.join(); torch.cuda.synchronize()
    bench.restart()                                  # frame 0 again: host[0] is frame 1 of the stream
    tsteps[0] = 0.0
    t0 = time.perf_counter()
    go(n // G, state, 0)
    bench.trk.join()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = float(state.cpu().numpy()[:, :, 6].mean())
    t1 = time.perf_counter()
    for k in range(32):
        HL.check(up_ctx.L.agt_upload(up_ctx.h, C.c_void_p(dbase + (k % NB) * chunk_bytes), C.c_void_p(hbase + ((k * G) % P) * B * H * W), chunk_bytes), "agt_upload")
    torch.cuda.synchronize()
    up_dt = (time.perf_counter() - t1) / (32 * G)
    bench.trk.join(); torch.cuda.synchronize()
    ctx_stream.__exit__(None, None, None)
    bench.restart()                                  # back on the default stream
    torch.cuda.synchronize()
    del host, devb
    return {"frames_per_s": round(B * n / dt, 1), "ms_per_step": round(dt / n * 1e3, 5), "accepted_frac": round(ok, 4),
            "upload_only_us_per_frame": round(up_dt * 1e6, 2), "upload_GBs": round(B * W * H / up_dt / 1e9, 2),
            "frames": n, "host_enqueue_us_per_frame": round(t_enq / n * 1e6, 2), "host_tracker_calls_us_per_frame": round(tsteps[0] / n * 1e6, 2),
            "note": "pinned host frames uploaded %d per copy on a copy stream (%d device buffers) while the tracker works on the previous chunk, "
                    "fused step at %d frames per launch; PCIe Gen5 x16 = 63 GB/s spec; wall time of %d frames including pipeline fill and drain"
                    % (G, NB * G, depth, n)}


def batch_extra(torch, D, HL, args, rank, dev):
    """BASELINE.json configs[2] side measurement: 64 independent 1280x720 streams per step."""
    a = argparse.Namespace(**vars(args))
    # A block = one detector-refresh interval (REDETECT = 600 frames, 20 s of video at 30 fps): the tracker joins exactly where the application
    # would -- before the corner set is refreshed -- and a join drains the pipeline (the pose role runs ~20 frames behind the LK role: 5 launches
    # of 4 frames with nothing beside them, ~0.35 ms).  Rounds 2-5 timed blocks of 256 steps, which carry that drain 2.3 x as often as the
    # workload does: 38.8-38.9 us per step against 37.3-37.4 on one box (gpurun_out/r6v); that figure stays on the line as `blocks_of_256`.
    a.steps, a.warmup, a.streams, a.render_frames = 256, 16, 64, min(args.render_frames, 8)
    wl = WORKLOADS["c3"]
    b = Bench(torch, wl, a, rank, 1, dev)
    b.trk.pipeline(16)                       # (the c3 workload's default group size)
    dts256, _, _, st256, _ = b.timed_blocks(D, 15)
    med256, p10_256, _ = percentiles(dts256)
    ok256 = float(st256[:, :, HL.ST_OK].mean())
    del b
    a.steps = REDETECT
    b = Bench(torch, wl, a, rank, 1, dev)
    b.trk.pipeline(16)
    dts, st_w, st_f, st, _ = b.timed_blocks(D, 15)
    med, p10, p90 = percentiles(dts)
    spans = b.stage_spans_us(HL, 40)
    ab = algorithmic_bytes(b.W, b.H, b.npts)
    step_bytes = 64 * ab["frame"]
    pyr_gbs = 64 * ab["pyramid"] / (float(spans[0]) * 1e-6) / 1e9
    pyr2_gbs = 64 * ab["pyramid_two_pass"] / (float(spans[0]) * 1e-6) / 1e9
    ok = float(st[:, :, HL.ST_OK].mean())
    # EVERY distinct rendered stream (4 seeds) against its own CPU chain, and every copy (stream b shows
    # seed b % 4 at its own HBM address) bitwise equal to the stream it copies -- a cross-stream mix-up cannot hide
    rec = np.concatenate([st_w[:a.warmup], st_f])
    pose_err = None
    if not args.no_cpu_baseline:
        gaps = [pose_gap_vs_cpu_chain(b.seqs[q], b.rendered[:, q], rec[:, q], b.NF, 25) for q in range(b.nseq)]
        pose_err = {"max_l2_drvec": max(g["max_l2_drvec"] for g in gaps), "max_l2_dtvec": max(g["max_l2_dtvec"] for g in gaps),
                    "streams_compared": b.nseq, "frames_each": gaps[0]["frames"], "tolerance": 1e-4}
    copies_ok = all(np.array_equal(rec[:, q].view(np.uint64), rec[:, q % b.nseq].view(np.uint64)) for q in range(64))
    del b
    torch.cuda.empty_cache()
    return {"workload": wl["label"] % 64 + " (span_us: serial pass of the stage kernels)", "frames_per_s": round(64 * a.steps / med, 1),
            "ms_per_step": round(med / a.steps * 1e3, 4), "ms_per_step_p10": round(p10 / a.steps * 1e3, 4), "ms_per_step_p90": round(p90 / a.steps * 1e3, 4),
            "whole_step_algorithmic_GBs": round(step_bytes / (med / a.steps) / 1e9, 1), "whole_step_frac_of_8TBs": round(step_bytes / (med / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "span_us": {"pyramid": round(float(spans[0]), 2), "lk": round(float(spans[1]), 2), "pnp": round(float(spans[2]), 2)},
            "pyr_down_algorithmic_GBs": round(pyr_gbs, 1), "pyr_down_frac_of_8TBs": round(pyr_gbs / HBM_PEAK_GBS, 4),
            "pyr_down_bytes_note": "W*H*1.3125 per frame (SURVEY 8d); on the bytes of two single-level passes (level 1 read again: rounds 3-4) that would be %.1f GB/s = %.4f of 8 TB/s"
                                   % (pyr2_gbs, pyr2_gbs / HBM_PEAK_GBS),
            "steps_per_block": a.steps, "blocks": 15,
            "blocks_of_256": {"ms_per_step": round(med256 / 256 * 1e3, 4), "ms_per_step_p10": round(p10_256 / 256 * 1e3, 4),
                              "whole_step_frac_of_8TBs": round(step_bytes / (med256 / 256) / 1e9 / HBM_PEAK_GBS, 4), "accepted_frac": round(ok256, 4)},
            "accepted_frac": round(ok, 4), "pose_err_vs_cpu": pose_err, "copies_bitwise_equal_to_their_seed_stream": bool(copies_ok)}


def pairs_extra(args):
    """BASELINE.json configs[2] exactly as SURVEY.md 8d states it, on the driver-timed line: 64 COLD 1280x720 frame
    pairs per step, both pyramids built, 169,638,912 algorithmic bytes per batch, 4 rotated batches (450 MiB of frames) -- the
    c3pairs workload's own measurement (bench_pairs.py), 15 blocks of 1,024 steps (0.7 s of GPU time; a block's fill, drain and gather
    cost 1-3 % of a 256-step block: 45.5 against 44.2 us per step at 256 / 4,096 steps on one box), run as a CHILD PROCESS of this command: it
    pipelines its batches over four HIP streams and wants the process's four hardware queues to itself; in this process, whose
    64-stream extra has created three library streams by now (or would create them afterwards), either measurement slowed the
    other by 1.5-1.8x (82 instead of 56 us per step; 87-91 instead of 48 the other way round).  Its CPU baseline runs with
    --workload c3pairs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AGT_DIST_BACKEND")}
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c3pairs", "--steps", "1024", "--warmup", "32", "--blocks", "15",
           "--render-frames", str(args.render_frames), "--no-cpu-baseline"]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not line:
            return {"error": "child exited %d: %s" % (p.returncode, p.stderr[-300:])}
        out = json.loads(line[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}
    keep = ("metric", "value", "unit", "steps", "ms_per_step", "config", "timing", "roofline", "max_abs_pose_err_vs_truth", "tracked_frac")
    res = {k: out[k] for k in keep}
    res["note"] = "child process of this command (python bench.py --workload c3pairs --steps 1024 --warmup 32 --blocks 15 --no-cpu-baseline)"
    return res


def c5_extra(args):
    """BASELINE.json configs[4] on the driver-timed line (c5 was builder-run only): 1280x720, 60 tags / 240 corners,
    LK + PnP + dense photometric refinement (61,440 samples, 5 Gauss-Newton iterations) + corner re-seed per frame -- the c5 workload's
    own default measurement (15 blocks of 400 steps), as a child process of this command (its own context and streams, like pairs64_hbm)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AGT_DIST_BACKEND")}
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c5", "--no-cpu-baseline"]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode or not line:
            return {"error": "child exited %d: %s" % (p.returncode, p.stderr[-300:])}
        out = json.loads(line[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}
    keep = ("metric", "value", "unit", "steps", "ms_per_step", "config", "timing", "roofline", "accepted_frac", "refined_frac", "tracked_corners_mean", "max_abs_pose_err_vs_truth")
    res = {k: out[k] for k in keep if k in out}
    res["note"] = "child process of this command (python bench.py --workload c5 --no-cpu-baseline)"
    return res


_native = []


def cpu_available():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_quota():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None: the GPU boxes list every core of the host in
    the affinity mask but grant a job about 16 cores' worth of time, which is why the thread curve collapses past 16 threads there"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def cpu_thread_counts():
    """1, 4, 16, 64, ... up to every core available to the process (SURVEY 8d's os.cpu_count() leg; the last entry)"""
    top = max(1, cpu_available())
    out = [c for c in (1, 4, 16, 64) if c < top]
    return out + [top] if top > 1 else [1]


def native_oracle():
    """the CPU oracle, rebuilt -O3 -march=native on this host the first time it is needed (cpu_baseline legs only)"""
    from oracle import cvoracle as cvo
    if not _native:
        _native.append(cvo.select_native() if cvo._lib is None else cvo.BUILD_FLAGS)
        cvo.build()
    return cvo, _native[0]


def live_latency(bench, n=300):
    """The reference's own use: ONE camera, one frame at a time -- hand the frame over, wait for its pose (step + join +
    synchronize per frame, one frame per launch: a pyramid launch, then LK + PnP chained in one launch).  Median / p90 of the
    host-observed time from the call to the finished record, frames resident in HBM."""
    torch = bench.torch
    bench.trk.pipeline(1)
    bench.restart()
    out = torch.zeros((bench.B, 16), dtype=torch.float64, device=bench.dev)
    ts = []
    for k in range(n + 20):
        f = bench.ring[(k + 1) % bench.ring_slots]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.trk.step(f, out); bench.trk.join(); torch.cuda.synchronize()
        if k >= 20:
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    ok = float(out.cpu().numpy()[:, 6].mean())
    return {"median": round(float(np.median(ts)), 1), "p90": round(float(np.percentile(ts, 90)), 1), "frames": n, "last_accepted": ok,
            "note": "step + join + synchronize per frame at one frame per launch (2 launches per frame)"}


def drop_in_latency(bench, n=200):
    """the class a reference user instantiates (detect_pose.py:57) on the device-resident path --
    PoseDetector(backend="stream")._detect_and_get_pose(frame), one HOST frame per call (agt_track_host_frame: the pinned frame read by the
    pyramid pass / gray kernel itself -> chained LK|PnP launch -> record polled in host-mapped memory).  Median / p90 of the host-observed call time over n tracked frames;
    gray (0.9 MB) and BGR (2.8 MB) frames, from the detector's pinned frame buffer and from an ordinary numpy array (one
    more host copy).  The first frame is seeded by a detector answer (ground truth), then the detector is removed (LK path)."""
    import json as js, logging, tempfile
    from accurate_aprilgroup_tracking_amd import formats
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    sq = bench.seqs[0]
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "april_group.json"), "w").write(js.dumps(sq.group))

    class Det(PoseDetector):
        DIRPATH = tmp
    log = logging.getLogger("bench"); log.setLevel(logging.CRITICAL)
    tag_ids = [int(t) for t in sq.group["tags"].keys()]
    frames = bench.rendered[:, 0]
    out = {}
    for name, color, pinned in (("gray_pinned", False, True), ("gray_pageable", False, False), ("bgr_pinned", True, True), ("bgr_pageable", True, False)):
        first = [formats.make_detection(t, c) for t, c in zip(tag_ids, sq.corners(0).reshape(-1, 4, 2))]
        det = Det(log, sq.K, None, True, detector=lambda gray: first, backend="stream")
        shape = frames[0].shape + ((3,) if color else ())
        buf = det.frame_buffer(shape) if pinned else np.empty(shape, np.uint8)
        srcs = [np.ascontiguousarray(np.stack([f, f, f], -1)) if color else f for f in frames]
        ts = []
        for k in range(n + 10):
            i = pingpong(k, bench.NF)
            if pinned:
                np.copyto(buf, srcs[i]); frame = buf          # (the capture's write into the buffer: not part of the call)
            else:
                frame = srcs[i]
            t0 = time.perf_counter()
            det._detect_and_get_pose(frame)
            dt = time.perf_counter() - t0
            if k == 0:
                det.detector = None
            if k >= 10:
                ts.append(dt)
        ts = np.array(ts) * 1e6
        ok = det.last_error is not None and det.last_error < 2
        out[name] = {"median": round(float(np.median(ts)), 1), "p90": round(float(np.percentile(ts, 90)), 1), "last_accepted": bool(ok)}
        del det
    out["note"] = ("PoseDetector(backend='stream')._detect_and_get_pose(host frame) at %dx%d, LK path, one frame per call: pyramid pass reading the pinned frame "
                   "over PCIe [BGR: gray kernel reading it] + chained LK|PnP launch, record polled in host-mapped memory; *_pageable adds the host "
                   "copy into the pinned staging buffer"
                   % (bench.W, bench.H))
    return out


def per_call_latency(bench):
    """INTEGRATION.md section 1 ("smallest change"): the synchronous numpy-in / numpy-out cv_hip calls a maintainer swaps in
    for cv2, next to the same calls answered by the CPU oracle on this host (median of 60 calls each, us)."""
    from accurate_aprilgroup_tracking_amd import cv_hip
    cvo = native_oracle()[0]
    sq = bench.seqs[0]
    a, b, pts = sq.frame(0), sq.frame(1), sq.corners(0)
    obj = sq.obj.astype(np.float32)
    nx = sq.corners(1)

    def med(fn, n=60):
        fn(); fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return round(float(np.median(ts)) * 1e6, 1)
    r0, t0_ = sq.rvecs[0].copy(), sq.tvecs[0].copy()
    # the C-ABI calls alone (arguments prepared once: what a compiled host pays; the cv_hip figures add ~10 us of numpy / ctypes marshalling)
    import ctypes as C
    from accurate_aprilgroup_tracking_amd import hiplib as HL
    npts = int(obj.shape[0])                 # (the sequence's corner count: ADVICE r5 -- not a literal 48)
    ctx = cv_hip._geom_context(npts)
    Kh = np.ascontiguousarray(sq.K.reshape(-1)); imgf = np.ascontiguousarray(nx, np.float32).reshape(npts, 2)
    pose = np.zeros(6); inf = np.zeros(4, np.int32); ptsf = np.empty((npts, 2), np.float32)
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    a_solve = (ctx.h, vp(obj), vp(imgf), HL.F32, npts, vp(Kh), None, 0, vp(pose), 1, vp(inf), None)
    a_init = (ctx.h, vp(obj), vp(imgf), HL.F32, npts, vp(Kh), None, 0, vp(pose), 0, vp(inf), None)
    a_proj = (ctx.h, vp(obj), HL.F32, npts, vp(pose), vp(Kh), None, 0, vp(ptsf), None)

    def c_solve(args):
        pose[:3] = r0; pose[3:] = t0_
        return ctx.L.agt_solve_pnp_host(*args)

    def checked(name, fn, solve):
        """one call whose return code and (solves) info[OK] are looked at before anything is timed: a call that fails at once would read as a
        very fast one"""
        rc = fn()
        if rc != 0 or (solve and inf[HL.INFO_OK] != 1):
            return {"error": "%s: rc %d, info %s" % (name, rc, inf.tolist())}
        return med(fn)
    with ctx.lock:
        ctx.use_current_stream()
        c_abi = {"agt_solve_pnp_host_guess_N%d" % npts: checked("agt_solve_pnp_host (guess)", lambda: c_solve(a_solve), True),
                 "agt_solve_pnp_host_noguess_N%d" % npts: checked("agt_solve_pnp_host (no guess)", lambda: c_solve(a_init), True),
                 "agt_project_points_host_N%d" % npts: checked("agt_project_points_host", lambda: ctx.L.agt_project_points_host(*a_proj), False)}
    out = {
        "c_abi_call_only": c_abi,
        "solvePnP_guess_N48": {"hip": med(lambda: cv_hip.solvePnP(obj, nx, sq.K, None, r0.copy(), t0_.copy(), True)),
                               "cpu_oracle": med(lambda: cvo.solvePnP(obj, nx, sq.K, None, r0.copy(), t0_.copy(), True))},
        "solvePnP_noguess_N48": {"hip": med(lambda: cv_hip.solvePnP(obj, nx, sq.K, None)),
                                 "cpu_oracle": med(lambda: cvo.solvePnP(obj, nx, sq.K, None))},
        "projectPoints_N48": {"hip": med(lambda: cv_hip.projectPoints(sq.obj, r0, t0_, sq.K, None)),
                              "cpu_oracle": med(lambda: cvo.projectPoints(sq.obj, r0, t0_, sq.K, None))},
        "calcOpticalFlowPyrLK_%dx%d_N48" % (bench.W, bench.H): {
            "hip": med(lambda: cv_hip.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2), 30),
            "cpu_oracle": med(lambda: cvo.calcOpticalFlowPyrLK(a, b, pts, maxLevel=2), 30)},
        "note": "one synchronous call: host arrays in, kernel(s), host arrays out; the device-resident StreamTracker is the throughput path",
    }
    return out


def cpu_baseline(seq, frames, gpu_state, Wm, K, NF):
    """Reference CPU path stand-in ("port"): the oracle's cvo_track_frame (full padded pyramids, full-frame Scharr image
    per level, per-point LK, FP64 LM; OpenCV's dataflow), compiled -O3 -march=native ON THIS HOST, on a bounded sample of the same
    stream: ~10 s on ONE core, ~5 s with the full-frame passes and the point loop threaded.  Plus the pose difference HIP
    vs the CPU chain on identical frames."""
    import logging, tempfile
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd import hiplib as HL
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    cvo, flags = native_oracle()
    H_, W_ = frames[0].shape
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

    def threaded(nthr, seconds):
        pyr2 = cvo.Pyramid(frames[0]); pts2 = seq.corners(0)
        r2, t2 = seq.rvecs[0].copy(), seq.tvecs[0].copy()
        n2 = 0; t1 = time.perf_counter()
        while True:
            k = pingpong(n2 + 1, NF)
            pyr2, pts2, _, _, _, r2, t2 = cvo.track_frame(pyr2, frames[k], pts2, seq.obj, seq.K, None, r2, t2, nthreads=nthr)
            n2 += 1
            if time.perf_counter() - t1 > seconds or n2 >= 8000:
                break
        return n2, time.perf_counter() - t1
    # "at 1 thread and at os.cpu_count() threads" -- and the steps between, so that the plateau is visible (one stream is
    # 48 corners and three image levels: pyrDown / Scharr in bands of rows, LK over points; the threads past ~16 have nothing to do)
    curve = [{"cores": 1, "value": round(n / dt, 2)}]
    for nthr in cpu_thread_counts()[1:]:
        n2, dt2 = threaded(nthr, 2.5)
        curve.append({"cores": nthr, "value": round(n2 / dt2, 2), "sample": "%d frames, %.1f s" % (n2, dt2)})
    top = curve[-1]
    cpu = {"value": round(n / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same %dx%d stream, oracle cvo_track_frame (pyramid+Scharr+LK+LM), 1 thread, %.1f s; host has %d cores, %d available to this process"
                     % (n, W_, H_, dt, os.cpu_count(), cpu_available()),
           "cpu_model": cpu_model(), "build": flags,
           "all_cores": {"value": top["value"], "cores": top["cores"],
                         "sample": "%s; pyrDown / Scharr in %d bands of rows, LK over points (OpenMP); every core available to the process" % (top.get("sample", ""), top["cores"])} if len(curve) > 1 else None,
           "threads_curve": curve, "best_of_curve": max(curve, key=lambda e_: e_["value"]), "cgroup_cpu_quota_cores": cpu_quota()}
    return cpu, pose_gap_vs_cpu_chain(seq, frames, gpu_state, NF, 60)


def pose_gap_vs_cpu_chain(seq, frames, gpu_state, NF, max_frames):
    """parity of what was timed: the reference-validated state machine on the oracle backend (oracle LK with the tracker's
    sticky status + PoseDetector mirror) over the first frames of ONE stream against that stream's device records"""
    import logging, tempfile
    from oracle import cv2_shim
    from accurate_aprilgroup_tracking_amd import hiplib as HL
    from accurate_aprilgroup_tracking_amd.pose_detector import PoseDetector
    cvo, _ = native_oracle()
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "april_group.json"), "w").write(json.dumps(seq.group))

    class Det(PoseDetector):
        DIRPATH = tmp
    log = logging.getLogger("bench"); log.setLevel(logging.CRITICAL)
    det = Det(log, seq.K, None, True, cv=cv2_shim.make_cv2())
    obj32 = seq.obj.astype(np.float32)
    npts = obj32.shape[0]
    pyr = cvo.Pyramid(frames[0]); pts = seq.corners(0)
    alive = np.ones(npts, bool)
    nchk = min(len(gpu_state), max_frames)
    dr = dtv = 0.0
    for i in range(nchk):
        npyr = cvo.Pyramid(frames[pingpong(i + 1, NF)])
        nx, status, _ = cvo.calcOpticalFlowPyrLK(pyr, npyr, pts, maxLevel=2)
        nx = nx.reshape(-1, 2); status = status.ravel().astype(bool)
        nx[~alive] = pts[~alive]; alive &= status              # the tracker's sticky status
        il = [nx[j].reshape(1, 1, 2) for j in range(npts) if alive[j]]
        ol = [obj32[j].reshape(1, 3) for j in range(npts) if alive[j]]
        det._estimate_pose(il if len(il) >= 8 else [], ol if len(il) >= 8 else [])
        if det.last_pose[0] is not None and gpu_state[i, HL.ST_OK]:
            dr = max(dr, float(np.linalg.norm(gpu_state[i, :3] - det.last_pose[0].ravel())))
            dtv = max(dtv, float(np.linalg.norm(gpu_state[i, 3:6] - det.last_pose[1].ravel().astype(np.float64))))
        pts = nx.astype(np.float32); pyr = npyr
    return {"max_l2_drvec": dr, "max_l2_dtvec": dtv, "frames": nchk, "tolerance": 1e-4}


if __name__ == "__main__":
    main()
