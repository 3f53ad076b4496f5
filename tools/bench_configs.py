#!/usr/bin/env python3
"""Secondary measurements on ONE MI355X for the other BASELINE.json configs (the headline is bench.py); the runners live in
deepmerge_amd/workload.py (bench.py reports them as `extras`), this is their command line:
  config 3  ViT-B/16 pair encoder, 224x224x3, 128 pairs/step, train step
  config 4  ExtractFeatures pipeline on a 4096x4096x4 tile: patch pyramid gather -> v3 [6,4,2] eval -> segment mean -> edge simi
  config 5  (single-GPU part) v3 [6,4,2], 4 scales x 4 ch, 120 pairs/step, train step
Prints one JSON line per config.   python tools/bench_configs.py [3] [4] [4r] [5] [5g]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmerge_amd.workload import config3, config4, config4r, config5, voronoi_raster  # noqa: E402,F401

if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    if "3" in which: print(json.dumps(config3()), flush=True)
    if "5" in which: print(json.dumps(config5()), flush=True)
    if "5g" in which: print(json.dumps(config5(graph=True)), flush=True)
    if "4" in which: print(json.dumps(config4()), flush=True)
    if "4r" in which: config4r()
