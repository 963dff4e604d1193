#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small tracked files under profiles/.

    python tools/summarize_prof.py <tag> --stats <dir> [--fetch <dir>] [--write <dir>] [--sq <dir>]

Writes profiles/<tag>_kernel_stats.csv (top kernels) and profiles/<tag>_pmc.json with, per kernel:
mean counters, HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 bytes (the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced
reads), effective clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD * 256 CU) / (GRBM_GUI_ACTIVE / 8)."""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name[:80]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[k]["_dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--keep", default="gram,gemm,symeig,deim,project,spmm,solve,reduce,scale")
    a = ap.parse_args()
    keep = a.keep.split(",")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    if a.stats:
        rows = []
        for f in glob.glob(os.path.join(a.stats, "*", "*_kernel_stats.csv")):
            for r in csv.DictReader(open(f)):
                rows.append([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                             r["MinNs"], r["MaxNs"]])
        with open(os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.csv"), "w") as fp:
            w = csv.writer(fp)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            w.writerows(rows[:25])
    pmc = collections.defaultdict(dict)
    for d in (a.fetch, a.write, a.sq):
        if d:
            for k, cs in counters(d).items():
                if any(s in k for s in keep):
                    pmc[k].update(cs)
    for k, cs in pmc.items():
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            cs["hbm_bytes_per_launch"] = 2.0 * cs.get("FETCH_SIZE", 0.0) * 1024 + cs.get("WRITE_SIZE", 0.0) * 1024
        if "GRBM_GUI_ACTIVE" in cs:
            cyc = cs["GRBM_GUI_ACTIVE"] / 8.0
            cs["effective_clock_GHz"] = cyc / cs["_dur_ms"] / 1e6
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
                cs["mfma_util"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
    with open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc.json"), "w") as fp:
        json.dump(pmc, fp, indent=1, sort_keys=True)
    gram = [cs["hbm_bytes_per_launch"] for k, cs in pmc.items() if k.startswith("gram128_") and "hbm_bytes_per_launch" in cs]
    tfile = os.path.join(ROOT, "profiles", "gram_traffic.json")
    if gram and "_bench_" in a.tag:   # only the headline workload's passes describe the bench line's Gram
        old = json.load(open(tfile)) if os.path.exists(tfile) else {}
        rec = dict(hbm_bytes_per_gram=sum(gram), source=f"profiles/{a.tag}_pmc.json",
                   workload="pod_1000000x512 on 1 GPU (bench.py default: pipeline mode, the Gram on its 224-CU stream, two "
                            "launches, unpaced)",
                   method="2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate --pmc passes, summed over the launches of one Gram")
        if "whole_chip" in old:
            rec["whole_chip"] = old["whole_chip"]
        with open(tfile, "w") as fp:
            json.dump(rec, fp, indent=1)
    if gram and "_benchlat_" in a.tag:   # bench.py --mode latency: the Gram of a single POD, all 256 CUs (one launch, paced)
        rec = json.load(open(tfile)) if os.path.exists(tfile) else {}
        rec["whole_chip"] = dict(hbm_bytes_per_gram=sum(gram), source=f"profiles/{a.tag}_pmc.json",
                                 kernels=sorted(k for k in pmc if k.startswith("gram128_")),
                                 workload="pod_1000000x512, bench.py --mode latency (one POD after the other on the whole chip)")
        with open(tfile, "w") as fp:
            json.dump(rec, fp, indent=1)
    proj = [cs for k, cs in pmc.items() if k.startswith("project_fused_kernel<5, false>") and "hbm_bytes_per_launch" in cs]
    if proj and "_c5sweep_" in a.tag:  # the projection launch of the direct sweep (32 value vectors, r = 80, N = 1e5)
        with open(os.path.join(ROOT, "profiles", "project_traffic.json"), "w") as fp:
            json.dump(dict(hbm_bytes_per_launch=proj[0]["hbm_bytes_per_launch"], source=f"profiles/{a.tag}_pmc.json",
                           workload="project_fused_kernel<5,false>, 32 value vectors on the pentadiagonal pattern, N = 1e5, r = 80 "
                                    "(tools/bench_configs.py c5sweep)",
                           method="2*FETCH_SIZE*1024 + WRITE_SIZE*1024, separate --pmc passes"), fp, indent=1)
    for k, cs in pmc.items():
        print(k, {c: (round(v, 4) if v < 1e4 else f"{v:.4g}") for c, v in cs.items()
                  if c in ("hbm_bytes_per_launch", "effective_clock_GHz", "mfma_util", "_dur_ms")})


if __name__ == "__main__":
    main()
