#!/usr/bin/env python3
"""oracle/probe/make_hw_tables.py -- TEST INFRASTRUCTURE.  Packs what oracle/_ref/hw_probe measured on the MI355X
(gpurun_out/hw/rsq_delta.bin, sqrt_delta.bin: int8[2^24] each, index = exponent parity << 23 | mantissa, value =
bits(v_rsq_f32(x)) - bits(1.0f / sqrtf(x)) resp. bits(v_sqrt_f32(x)) - bits(sqrtf(x)) for x in [1, 4)) into
oracle/hw_tables.bin.z (zlib).  The probe's report (exhaustive scaling check, min/max/med3 truth table) is kept as
oracle/hw_probe_report.txt.

    gpurun -- 'mkdir -p gpurun_out/hw && oracle/_ref/hw_probe gpurun_out/hw'  &&  python oracle/probe/make_hw_tables.py
"""
import os
import shutil
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "hw")
r = np.fromfile(os.path.join(src, "rsq_delta.bin"), np.int8)
s = np.fromfile(os.path.join(src, "sqrt_delta.bin"), np.int8)
assert r.size == s.size == 1 << 24
blob = zlib.compress(r.tobytes() + s.tobytes(), 9)
open(os.path.join(ROOT, "oracle", "hw_tables.bin.z"), "wb").write(blob)
shutil.copy(os.path.join(src, "report.txt"), os.path.join(ROOT, "oracle", "hw_probe_report.txt"))
print(f"rsq deltas {dict(zip(*np.unique(r, return_counts=True)))}, sqrt deltas {dict(zip(*np.unique(s, return_counts=True)))}, {len(blob)} bytes")
