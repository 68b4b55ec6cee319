#!/usr/bin/env python3
"""Where one conv_frame_f16s launch spends its time: tagged in-kernel stamps (wave 0 of every block) of the -DFR_STAMPS
build (conv_mfma.hip).  Runs bench.py in-process for one step and reads the stamps of the VSP_STAMP_FRAME-th launch.
usage (tools/build_stamps.sh first; GPU box): VSP_LIB_PATH=build/frstamps/libvispeech_hip.so VSP_STAMP_FRAME=<n> python tools/stamps_frame.py [bench args]
Tags: 1 start | 2 requests out | 3 first window in (wait + barrier) | 4 converted | per step: 10 top, 11 waited, 12 barrier,
13 copies issued, 14 fragment reads + MFMAs issued, 15 next window converted | 30 loop done | 31 stores issued | 32 retired."""
import ctypes as C
import contextlib
import io
import os
import runpy
import sys
from collections import defaultdict

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
args = sys.argv[1:] or ["--workload", "C2", "--batch", "1"]
sys.argv = ["bench.py"] + args + ["--steps", "1", "--warmup", "0", "--profile-steps", "0", "--no-cpu-baseline"]
with contextlib.redirect_stdout(io.StringIO()):
    try:
        runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
from vispeech_amd import _lib   # noqa: E402
lib = _lib.lib()
fn = lib.vsp_debug_stamps_frame
fn.restype = C.c_int
NS, NSTAMP = 64, 512
buf = np.zeros((NS, NSTAMP), dtype=np.uint64)
n = fn(buf.ctypes.data_as(C.c_void_p), NS, 1)
print(f"conv_frame_f16s launch #{os.environ.get('VSP_STAMP_FRAME')}: {n} sampled blocks")
NAMES = {(1, 5): "epilogue operands requested", (5, 6): "weight copies issued", (6, 2): "window copies issued", (2, 3): "first window: wait + barrier", (3, 4): "first window: convert",
         (4, 10): "-", (10, 11): "counted wait (vmcnt)", (11, 12): "barrier", (12, 13): "copies issued",
         (13, 14): "fragment reads + MFMAs issued", (14, 15): "next window: convert", (14, 10): "-", (15, 10): "-",
         (14, 30): "-", (15, 30): "-", (30, 31): "epilogue: loads + stores issued", (31, 32): "stores retired"}
tot = []
per = defaultdict(list)
steps = []
for s in buf[:n]:
    tags = (s >> np.uint64(56)).astype(np.int64)
    t = (s & np.uint64(0x00ffffffffffffff)).astype(np.int64)
    k = int((tags > 0).sum())
    if k < 4:
        continue
    acc = defaultdict(float)
    for i in range(1, k):
        acc[NAMES.get((int(tags[i - 1]), int(tags[i])), f"{tags[i-1]}->{tags[i]}")] += (t[i] - t[i - 1]) / 100.0
    for kk, v in acc.items():
        per[kk].append(v)
    tot.append((t[k - 1] - t[0]) / 100.0)
    steps.append(int((tags == 10).sum()))
print(f"block lifetime: median {np.median(tot):.2f} us  (min {np.min(tot):.2f}, max {np.max(tot):.2f}), {len(tot)} blocks, {int(np.median(steps))} steps")
for kk, v in sorted(per.items(), key=lambda kv: -np.median(kv[1])):
    print(f"  {kk:38s} {np.median(v):8.2f} us  {100 * np.median(v) / np.median(tot):5.1f} %")
# one block, step by step
s = buf[0]
tags = (s >> np.uint64(56)).astype(np.int64)
t = (s & np.uint64(0x00ffffffffffffff)).astype(np.int64)
k = int((tags > 0).sum())
line = []
for i in range(k):
    line.append(f"{int(tags[i])}@{(t[i] - t[0]) / 100.0:.2f}")
print("block 0:", " ".join(line))
