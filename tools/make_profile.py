"""Turn rocprofv3 output directories into the committed profile summary (profiles/*.md + *_traffic.json).

    python tools/make_profile.py --tag r01_onepass --stats DIR --fetch DIR --write DIR --bench-line FILE --cmd "..."

DIRs are what `rocprofv3 -d DIR` wrote (one kernel-trace/--stats run, one --pmc FETCH_SIZE run, one
--pmc WRITE_SIZE run: counters are collected in their own passes).  FETCH_SIZE / WRITE_SIZE are in KiB
(MI355X_MICROARCH.md, HBM section); wide coalesced reads are under-counted by 2x on gfx950, so the read
traffic bench.py reports is 2 x FETCH_SIZE.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def counter_means(d, name):
    per = defaultdict(list)
    with open(find(d, "*counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name:
                per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--bench-line", required=True)
    ap.add_argument("--cmd", required=True)
    ap.add_argument("--workload", default="cfg3 1024 images seed 3")
    a = ap.parse_args()

    bench = json.loads(open(a.bench_line).read().strip().splitlines()[-1])
    stats = open(find(a.stats, "*kernel_stats.csv")).read().strip().splitlines()
    fetch, nf = counter_means(a.fetch, "FETCH_SIZE")
    write, nw = counter_means(a.write, "WRITE_SIZE")

    md = [f"# profiles/{a.tag}.md -- rocprofv3 evidence", "",
          "Commands (MI355X box, ROCm 7.2, `cd /tmp && export TMPDIR=/tmp` first; counters in their own passes):", "",
          f"    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- {a.cmd}",
          f"    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- {a.cmd}",
          f"    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- {a.cmd}", "",
          f"bench.py line of the --stats run: value {bench['value']} {bench['unit']}, ms_per_step {bench['ms_per_step']}, "
          f"dominant kernel {bench['roofline']['kernel']} {bench['roofline']['kernel_ms']} ms (HIP events), kernels_ms {bench['kernels_ms']}", "",
          "## kernel stats (all decodes of the run: warm-up + timed graph replays + event-timed launches)", "", "```"]
    md += [ln for ln in stats if ln.startswith('"Name"') or "pjd_k" in ln or "rocclr" in ln]
    md += ["```", "", "## HBM traffic per launch (mean over the launches of the PMC runs; KiB)", "",
           "| kernel | FETCH_SIZE KiB (raw) | read MiB = 2 x FETCH_SIZE (gfx950 correction) | WRITE_SIZE KiB | launches |", "|---|---|---|---|---|"]
    per_kernel = {}
    for k in sorted(set(fetch) | set(write)):
        per_kernel[k] = {"fetch_kib_raw": fetch.get(k), "write_kib": write.get(k)}
        md.append(f"| `{k[:60]}` | {fetch.get(k, 0):.0f} | {2 * fetch.get(k, 0) / 1024:.1f} | {write.get(k, 0):.0f} | {nf.get(k, 0)} |")
    md.append("")
    open(os.path.join(ROOT, "profiles", a.tag + ".md"), "w").write("\n".join(md))
    json.dump({"workload": a.workload, "source": f"profiles/{a.tag}.md", "per_kernel": per_kernel},
              open(os.path.join(ROOT, "profiles", a.tag + "_traffic.json"), "w"), indent=1)
    print("wrote", a.tag)


if __name__ == "__main__":
    main()
