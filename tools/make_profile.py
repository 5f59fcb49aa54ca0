"""Turn the rocprofv3 output of tools/r2_pmc.sh into the committed evidence: profiles/<tag>.md + profiles/<tag>_counters.json.

    python tools/make_profile.py --tag r02_cfg3 --dir gpurun_out/pmc_cfg3 --workload-key cfg3

--dir holds what tools/r2_pmc.sh wrote: stats/ (--kernel-trace --stats), sq1/ sq2/ sq3/ (SQ counters, eight per pass),
fetch/ (FETCH_SIZE), write/ (WRITE_SIZE) -- counters are collected in their own passes, never together with a trace of
the HIP/HSA API.  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are
in KiB and wide coalesced reads are under-counted by 2x on gfx950, so read traffic = 2 x FETCH_SIZE.

Derived figures (per kernel, median over the launches of a pass):
    traffic_bytes    = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
    kernel cycles    = SQ_BUSY_CYCLES / 32          (the counter sums the 32 shader engines of the 8 XCDs)
    valu_issue_frac  = 4 x SQ_INSTS_VALU / (1024 SIMDs x kernel cycles)     a wave64 VALU instruction holds its SIMD 4 cycles
    lane_util        = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)   active lanes per issued VALU instruction
    wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES                         wave-cycles spent waiting on a counter (vm/lgkm/exp)
bench.py reads <tag>_counters.json for `roofline.traffic` and `roofline.valu_issue_frac`.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = {"pjd_k_huff_lanes": "huff_lanes", "pjd_k_idct_colour_lanes": "idct_colour", "pjd_k_lane_words": "lane_words",
         "pjd_k_build_tables": "build_tables", "pjd_k_lane_dc_local": "dc_scan (local)", "pjd_k_lane_dc_carry": "dc_scan (carry)",
         "pjd_k_huff_sequential": "huff_sequential", "pjd_k_idct_colour": "idct_colour (dense)"}


def counters(d):
    per = defaultdict(lambda: defaultdict(list))
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                          # the newest run only (gpurun merges into an existing directory)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    med = lambda v: sorted(v)[len(v) // 2]      # median: a run also holds one-picture launches (the bench decodes its bundled sample first)
    return {k: {c: med(v) for c, v in cs.items()} for k, cs in per.items()}, {k: max(len(v) for v in cs.values()) for k, cs in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--workload-key", default="cfg3")
    ap.add_argument("--note", default="")
    a = ap.parse_args()

    bench = None
    for ln in open(os.path.join(a.dir, "stats.log")).read().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            bench = json.loads(ln)
    stats_csv = sorted(glob.glob(os.path.join(a.dir, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    stats = open(stats_csv[-1]).read().strip().splitlines() if stats_csv else []
    C, N = {}, {}
    for sub in ("sq1", "sq2", "sq3", "fetch", "write"):
        c, n = counters(os.path.join(a.dir, sub))
        for k, v in c.items():
            C.setdefault(k, {}).update(v)
            N[k] = max(N.get(k, 0), n[k])
    cmd = "python3 bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10"
    md = [f"# profiles/{a.tag}.md -- rocprofv3 evidence, {a.workload_key}", "", a.note, "",
          "Commands (`tools/r2_pmc.sh`, MI355X box, ROCm 7.2, `cd /tmp && export TMPDIR=/tmp` first; one pass per line):", "",
          f"    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- {cmd}",
          f"    rocprofv3 --pmc <8 SQ counters> --kernel-trace --output-format csv -d <dir> -- {cmd}      (three passes)",
          f"    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- {cmd}",
          f"    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- {cmd}", ""]
    if bench:
        md += [f"bench.py line of the --stats run: value {bench['value']} {bench['unit']} ({bench['config']['workload']}), "
               f"kernels_ms (HIP events, launches alone on the stream) {bench['kernels_ms']}; "
               f"ECS {bench['config']['ecs_bytes_per_gpu']} B, {bench['config']['pixels_per_gpu']} pixels, "
               f"{bench['config']['huffman_lanes']} lanes of {bench['config'].get('sub_bytes', '?')} B.", ""]
    md += ["## kernel stats (every launch of the run: warm-up, graph replays, event-timed launches)", "", "```"]
    md += [ln for ln in stats if ln.startswith('"Name"') or "pjd_k" in ln or "rocclr" in ln]
    md += ["```", ""]

    per_kernel = {}
    alg = bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None
    md += ["## HBM traffic per launch (median over launches)", "",
           "| kernel | FETCH_SIZE KiB (raw) | read MB = 2 x FETCH (gfx950 correction) | WRITE_SIZE KiB | written MB | launches |", "|---|---|---|---|---|---|"]
    tot_r = tot_w = 0.0
    for k in sorted(C):
        v = C[k]
        if "FETCH_SIZE" not in v and "WRITE_SIZE" not in v:
            continue
        f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        rd, wr = 2 * f * 1024, w * 1024
        if k.startswith("pjd_k"):
            tot_r += rd
            tot_w += wr
        per_kernel.setdefault(k, {}).update({"fetch_kib_raw": f, "write_kib": w, "traffic_bytes": int(rd + wr)})
        md.append(f"| `{SHORT.get(k, k)[:40]}` | {f:.0f} | {rd / 1e6:.1f} | {w:.0f} | {wr / 1e6:.1f} | {N.get(k, 0)} |")
    if alg:
        md += ["", f"Decode kernels together: read {tot_r / 1e6:.0f} MB + written {tot_w / 1e6:.0f} MB = {(tot_r + tot_w) / 1e6:.0f} MB per batch against "
               f"{alg / 1e6:.0f} MB algorithmic (bitstreams read once + pictures written once): x{(tot_r + tot_w) / alg:.2f}"
               f" (x{(tot_r / 2 + tot_w) / alg:.2f} without the read correction)."]
    md += ["", "## SQ counters per launch (median)", "",
           "| kernel | VALU insts | SALU | LDS | VMEM rd | VMEM wr | branch | valu_issue_frac | lane_util | wait_frac | LDS bank-conflict cycles / LDS active |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(C):
        v = C[k]
        if "SQ_INSTS_VALU" not in v or not k.startswith("pjd_k"):
            continue
        cyc = v.get("SQ_BUSY_CYCLES", 0) / 32.0
        vif = 4 * v["SQ_INSTS_VALU"] / (1024 * cyc) if cyc else None
        lu = v.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * v["SQ_ACTIVE_INST_VALU"]) if v.get("SQ_ACTIVE_INST_VALU") else None
        wf = v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None
        bc = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else None
        per_kernel.setdefault(k, {}).update({"valu_issue_frac": round(vif, 4) if vif is not None else None,
                                             "lane_util": round(lu, 4) if lu is not None else None,
                                             "wait_frac": round(wf, 4) if wf is not None else None,
                                             "sq": {c: int(x) for c, x in v.items() if c.startswith("SQ_")}})
        fmt = lambda x: "-" if x is None else f"{x:.3f}"
        md.append(f"| `{SHORT.get(k, k)}` | {v['SQ_INSTS_VALU']:.3g} | {v.get('SQ_INSTS_SALU', 0):.3g} | {v.get('SQ_INSTS_LDS', 0):.3g} | "
                  f"{v.get('SQ_INSTS_VMEM_RD', 0):.3g} | {v.get('SQ_INSTS_VMEM_WR', 0):.3g} | {v.get('SQ_INSTS_BRANCH', 0):.3g} | "
                  f"{fmt(vif)} | {fmt(lu)} | {fmt(wf)} | {fmt(bc)} |")
    md += ["", "valu_issue_frac = 4 x SQ_INSTS_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32); lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
           "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", ""]
    open(os.path.join(ROOT, "profiles", a.tag + ".md"), "w").write("\n".join(md))
    # bench.py looks kernels up by the short names of its kernels_ms
    out = {"workload_key": a.workload_key, "source": f"profiles/{a.tag}.md",
           "per_kernel": {SHORT.get(k, k).replace(" (", "_").replace(")", ""): v for k, v in per_kernel.items()}}
    json.dump(out, open(os.path.join(ROOT, "profiles", a.tag + "_counters.json"), "w"), indent=1)
    print("wrote profiles/" + a.tag + ".md and _counters.json")


if __name__ == "__main__":
    main()
