#!/usr/bin/env python3
"""tools/single_slices.py [n_slices...] -- cfg 3 on a single-image engine (ngd_config.single_image) with the engine's own
slice count (0) and with given ones: ms of the accumulation phase (the ranges' formation + launches) per pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

for ks in [int(v) for v in sys.argv[1:]] or [0, 48, 64, 88, 112, 136, 160, 248]:
    with N.Engine(1000, 1_000_000, kernel="mfma", single_image=True, n_slices=ks) as e:
        e.synth_fill(3)
        ms = []
        for _ in range(5):
            e.run()
            ms.append(e.timing()["ms_accum"])
        print("n_slices %3d: accumulation %.2f ms per pass (min of %s), %.1f GB on the device"
              % (ks, min(ms), " ".join("%.1f" % x for x in ms), e.device_bytes() / 1e9), flush=True)
