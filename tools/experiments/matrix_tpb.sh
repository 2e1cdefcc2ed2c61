#!/bin/bash
# settled launch duration of the distance matrix (32000^2) for several tiles-per-workgroup settings (MORB_MATRIX_TPB)
R=$GRAFT_REPO_ROOT
for t in ${TPBS:-0 4 5 8 11 16 21 32 0}; do
  MORB_MATRIX_TPB=$t python3 - <<P
import os, sys, time
sys.path.insert(0, "$R")
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, rt
n = 32000
d = synth.descriptors(n, 777); qh = synth.perturbed_queries(d, 11)
dq = rt.DeviceBuffer(n * 32); dr = rt.DeviceBuffer(n * 32); dq.upload(qh); dr.upload(d)
dout = rt.DeviceBuffer(n * n * 2)
run = lambda: m.Matcher.hamming_matrix_device(dq.ptr, n, dr.ptr, n, dout.ptr, None)
for _ in range(250): run()
e0, e1 = rt.Event(), rt.Event()
e0.record(None)
for _ in range(100): run()
e1.record(None)
us = e0.elapsed_ms(e1) * 10
print("MORB_MATRIX_TPB=%s: %.1f us  %.3f of 8 TB/s" % ("$t", us, (2.0 * n * n + 64.0 * n) / us / 1e6 / 8.0))
P
done
