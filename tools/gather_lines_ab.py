#!/usr/bin/env python3
"""A/B on the C5 remap: the tiled gather staging 16-byte chunks (default) against whole 128-byte lines (SBA_GATHER_LINES=1), for
several LDS budgets and frames per block.  One child process per configuration (tools/gather_sweep.py's child).
Usage: python tools/gather_lines_ab.py [frames=256]"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
configs = [{}]
for budget in (8192, 12288, 16384, 24576, 32768):
    for fpb in (1, 2):
        configs.append({"SBA_GATHER_LINES": "1", "SBA_GATHER_LDS_BUDGET": str(budget), "SBA_GATHER_FPB": str(fpb)})
for cfg in configs:
    r = subprocess.run([sys.executable, os.path.join(HERE, "gather_sweep.py"), "--child", str(F)], env=dict(os.environ, **cfg),
                       capture_output=True, text=True, timeout=180)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    print(json.dumps(cfg), line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
