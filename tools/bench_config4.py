#!/usr/bin/env python3
"""python tools/bench_config4.py [--steps K] [--warmup W]: BASELINE.json configs[3] alone -- the same measurement bench.py
carries as the `config4` object of its default line (bench.config4_leg); equivalent to `python bench.py --config4`."""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--keyframes", type=int, default=256)
a = ap.parse_args()
from lineslam_amd import build  # noqa: E402
build.build()
print(json.dumps(bench.config4_leg(steps=a.steps, warmup=a.warmup, keyframes=a.keyframes)))
