#!/usr/bin/env python
"""profiles/pmc_<window>.json (what bench.py's roofline.traffic reads) from the two counter summaries of a PMC collection:

  python tools/pmc_json.py W12 gpurun_out/<tag>/pmc_W12_FETCH_SIZE.csv gpurun_out/<tag>/pmc_W12_WRITE_SIZE.csv profiles/pmc_W12.json [avg|min]

FETCH_SIZE / WRITE_SIZE are per-kernel averages in KiB (tools/rocpd_summary.py counters); FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE taken as is; both are printed beside the streaming kernels of known size of
the same run (k_calib_read: ntiles x 9216 B read, k_calib_write: nchunks x Dm^2 x 4 B written)."""
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def rows(path):
    return {r["Kernel"]: r for r in csv.DictReader(open(path))}


def pick(table, prefix, stat):
    for name, r in table.items():
        if name.startswith(prefix):
            return float(r[{"avg": "Average", "min": "Min"}[stat]])
    return None


def main():
    window, fetch_csv, write_csv, out = sys.argv[1:5]
    stat = sys.argv[5] if len(sys.argv) > 5 else "avg"
    from sos_slam_amd import synth
    win = synth.make_window(window)
    F, W = rows(fetch_csv), rows(write_csv)
    import subprocess
    import bench
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=bench.ROOT).stdout.strip() or None
    except OSError:
        commit = None
    commit = commit or __import__("os").environ.get("SOS_SOURCE_COMMIT")   # (the GPU box has no .git: tools/gpu_settle.sh passes it in)
    d = {"source_commit": commit, "kernel_sources_sha": bench.kernel_sources_sha(),
         "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate kernel-trace-only passes over tools/pmc_probe.py {window} "
                   f"(tools/collect_profiles.sh): {fetch_csv}, {write_csv}; statistic over the dispatches: {stat}",
         "window": window, "residuals": int(win.R),
         "corrections": "FETCH_SIZE (KiB) doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE (KiB) taken as is",
         "calibration": {"k_calib_read": {"FETCH_SIZE_KiB": pick(F, "k_calib_read", "avg")},
                         "k_calib_write": {"WRITE_SIZE_KiB": pick(W, "k_calib_write", "avg")}}}
    for key, prefix in (("k_linearize_fused", "void k_linearize2<true"), ("k_linearize_unfused", "void k_linearize2<false")):
        f, w = pick(F, prefix, "avg"), pick(W, prefix, stat)
        if f is None or w is None:
            continue
        traffic = int(round((2 * f + w) * 1024))
        d[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": traffic,
                  "traffic_bytes_per_residual": round(traffic / win.R, 1)}
    d["k_linearize_fused"]["algorithmic_bytes_per_launch_survey_8d"] = int(win.R) * 1088
    d["k_linearize_fused"]["min_bytes_per_launch_fused"] = int(win.R) * 568
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d["k_linearize_fused"]))


if __name__ == "__main__":
    main()
