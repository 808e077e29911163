#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the tracked summaries under profiles/.

   python tools/summarize_profiles.py gpurun_out/prof_r01 r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE means, raw and corrected) and
profiles/<tag>_summary.md.  HBM bytes follow MI355X_MICROARCH.md section HBM: the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read (16 B per lane),
so the read side is doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name: str) -> str:
    name = name.replace("sba::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    src, tag = Path(sys.argv[1]), sys.argv[2]
    out = ROOT / "profiles"
    out.mkdir(exist_ok=True)
    shutil.copy(src / "trace" / "trace_kernel_stats.csv", out / f"{tag}_kernel_stats.csv")
    stats = {short(r["Name"]): r for r in csv.DictReader(open(src / "trace" / "trace_kernel_stats.csv"))}
    pmc = collections.defaultdict(dict)
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(src / f"pmc_{kind}" / f"{kind}_counter_collection.csv")):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            pmc[k][counter] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
    bench = {}
    for nm in ("trace", "fetch", "write"):
        try:
            bench[nm] = json.loads((src / f"bench_{nm}.json").read_text().strip().splitlines()[-1])
        except Exception:
            pass
    summary = {"tag": tag, "kernels": {}}
    for k, d in pmc.items():
        f = d.get("FETCH_SIZE", {}).get("mean_KiB", 0.0)
        w = d.get("WRITE_SIZE", {}).get("mean_KiB", 0.0)
        summary["kernels"][k] = {
            "FETCH_SIZE_mean_KiB": f, "WRITE_SIZE_mean_KiB": w,
            "hbm_read_bytes_per_launch": 2.0 * f * 1024.0,      # gfx950: FETCH_SIZE = 1/2 of a wide streaming read
            "hbm_write_bytes_per_launch": w * 1024.0,
            "launches": d.get("FETCH_SIZE", {}).get("launches"),
            "avg_duration_ns": float(stats[k]["AverageNs"]) if k in stats else None,
            "calls_in_trace": int(stats[k]["Calls"]) if k in stats else None,
        }
    if "trace" in bench:
        summary["bench_line_under_trace"] = bench["trace"]
    (out / f"{tag}_pmc.json").write_text(json.dumps(summary, indent=1))
    (out / "pmc_latest.json").write_text(json.dumps(summary, indent=1))
    lines = [f"# rocprofv3 summary {tag}", "",
             "Command (tools/profile_round.sh): `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 "
             "--no-cpu-baseline`, then separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same command.", "",
             "| kernel | calls | avg us | min us | max us | FETCH_SIZE KiB | read bytes (x2, gfx950) | WRITE_SIZE KiB |",
             "|---|---|---|---|---|---|---|---|"]
    for k, r in stats.items():
        s = summary["kernels"].get(k, {})
        lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                     f"{float(r['MaxNs'])/1e3:.2f} | {s.get('FETCH_SIZE_mean_KiB', 0):.1f} | "
                     f"{s.get('hbm_read_bytes_per_launch', 0):.4g} | {s.get('WRITE_SIZE_mean_KiB', 0):.1f} |")
    if "trace" in bench:
        b = bench["trace"]
        rf = b["roofline"]
        lines += ["", f"bench.py under the trace: sweep kernel {rf['kernel_ms']*1e3:.1f} us by HIP events "
                      f"(rocprofv3 average above), algorithmic {rf['algorithmic_bytes_per_launch']/1e6:.1f} MB per launch, "
                      f"{rf['achieved']:.0f} GB/s = {rf['frac']*100:.1f} % of {rf['peak']:.0f} GB/s."]
    (out / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
