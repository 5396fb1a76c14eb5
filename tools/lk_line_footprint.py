#!/usr/bin/env python3
"""128-byte-line footprint of the LK tracker's tile loads on the c3pairs scene (VERDICT r4 #1d): how many bytes a launch must move in
whole cache lines (a) when no corner shares a line with another, (b) when the corners of one frame share every line they have in common
(a perfect L2 for the frame), against the bytes the tiles hold and the PMC figure (profiles/pmc_traffic.json, 2 x FETCH_SIZE = lines,
profiles/r05_fetch_size_calibration.json).  Tile geometry as agt_lk_rs_body.h's prologue: per level an I tile of 24 rows x 7 aligned
dwords at ((floor(p - 10) - 1) & ~3, floor(p - 10) - 1) and a J tile of 40 rows x 11 aligned dwords at ((floor(g - 10) - 9) & ~3, ...),
level pitch = width rounded up to 64 (level 0: the caller's pitch = width).  CPU only.

    python3 tools/lk_line_footprint.py [--pairs 64]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from accurate_aprilgroup_tracking_amd import synthetic as syn  # noqa: E402

LINE = 128
WIN, MARGIN, LEVELS = 21, 9, 3
IW, INDW = WIN + 3, (WIN + 3 + 6) // 4
JT, JNDW = WIN + 1 + 2 * MARGIN, (WIN + 1 + 2 * MARGIN + 6) // 4


def tile_lines(base, pitch, w, h, ax0, ty0, rows, ndw):
    """set of line addresses one tile touches (rows / columns clipped to the image as the border path would reflect inside it)"""
    out = set()
    for r in range(rows):
        y = min(max(ty0 + r, 0), h - 1)
        x0, x1 = max(ax0, 0), min(ax0 + 4 * ndw, w) - 1
        a0, a1 = base + y * pitch + x0, base + y * pitch + x1
        for ln in range(a0 // LINE, a1 // LINE + 1):
            out.add(ln)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--frames", type=int, default=9)
    args = ap.parse_args()
    W, H = 1280, 720
    nseq = 4
    seqs = [syn.Sequence(W, H, n_tags=12, n_frames=args.frames, seed=s, supersample=1, group_seed=0) for s in range(nseq)]     # (frames are rendered lazily: none here)
    lw = [W >> l if l == 0 else (W + (1 << l) - 1) >> l for l in range(LEVELS)]
    lh = [(H + (1 << l) - 1) >> l for l in range(LEVELS)]
    lp = [W] + [(w + 63) & ~63 for w in lw[1:]]
    halfw = (WIN - 1) * 0.5
    tot = dict(tile_bytes=0, lines_no_sharing=0, lines_shared_in_frame=0, lines_shared_in_tag=0, corners=0)
    per_level = [dict(no_sharing=0, shared_in_frame=0) for _ in range(LEVELS)]
    for b in range(args.pairs):
        sq, k = seqs[b % nseq], (b // nseq) % (args.frames - 1)
        pts = np.asarray(sq.corners(k), np.float32)
        frame_I = [set() for _ in range(LEVELS)]
        frame_J = [set() for _ in range(LEVELS)]
        tag_sets = {}
        for i, (px, py) in enumerate(pts):
            for l in range(LEVELS):
                s = np.float32(1.0) / np.float32(1 << l)
                ipx, ipy = int(np.floor(np.float32(px) * s - np.float32(halfw))), int(np.floor(np.float32(py) * s - np.float32(halfw)))
                # base addresses: every level / image of a pair is its own allocation; give each a line-aligned base far apart
                ti = tile_lines(0, lp[l], lw[l], lh[l], (ipx - 1) & ~3, ipy - 1, IW, INDW)
                tj = tile_lines(0, lp[l], lw[l], lh[l], (ipx - MARGIN) & ~3, ipy - MARGIN, JT, JNDW)
                n = len(ti) + len(tj)
                tot["lines_no_sharing"] += n
                per_level[l]["no_sharing"] += n
                frame_I[l] |= ti; frame_J[l] |= tj
                tg = tag_sets.setdefault((i // 4, l), [set(), set()])
                tg[0] |= ti; tg[1] |= tj
            tot["tile_bytes"] += LEVELS * (IW * IW + JT * JT)
            tot["corners"] += 1
        for l in range(LEVELS):
            n = len(frame_I[l]) + len(frame_J[l])
            tot["lines_shared_in_frame"] += n
            per_level[l]["shared_in_frame"] += n
        tot["lines_shared_in_tag"] += sum(len(a) + len(b_) for a, b_ in tag_sets.values())
    scale = 3072.0 / tot["corners"]
    out = {
        "scene": "c3pairs: %d pairs x %d corners, 1280x720, 3 levels, 21x21 window (synthetic.Sequence seeds 0..3, group_seed 0)" % (args.pairs, tot["corners"] // args.pairs),
        "per_3072_corner_launch_MB": {
            "bytes_the_tiles_hold (24x24 + 40x40 per level)": round(tot["tile_bytes"] * scale / 1e6, 2),
            "whole_lines_no_sharing": round(tot["lines_no_sharing"] * LINE * scale / 1e6, 2),
            "whole_lines_shared_within_a_tag (4 corners)": round(tot["lines_shared_in_tag"] * LINE * scale / 1e6, 2),
            "whole_lines_shared_within_a_frame (48 corners: the floor of any launch order)": round(tot["lines_shared_in_frame"] * LINE * scale / 1e6, 2),
        },
        "per_level_MB": [{k_: round(v * LINE * scale / 1e6, 2) for k_, v in d.items()} for d in per_level],
    }
    try:
        pm = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")))
        out["measured_2xFETCH_SIZE_MB"] = round(pm["lk_kernel<21,1,3> c3pairs"]["traffic_bytes_per_launch"] / 1e6, 2)
    except (OSError, KeyError):
        pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
