"""Drive tools/sync_sim.c over pictures of the cfg3 set: sync distances and re-sync rounds per speculation policy.

    python tools/sync_sim.py [n_pictures] [S]

Measurement tooling (CPU only; the oracle's scanner is used to get tables and the destuffed stream)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import synth  # noqa: E402
from oracle_lib import Port  # noqa: E402

SO = os.path.join(HERE, "libsyncsim.so")
SRC = os.path.join(HERE, "sync_sim.c")


def lib():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", SO, SRC], check=True)
    L = C.CDLL(SO)
    L.sim_open.restype = C.c_void_p
    L.sim_open.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 6 + [C.c_int, C.c_int]
    L.sim_close.argtypes = [C.c_void_p]
    L.sim_policy.argtypes = [C.c_void_p, C.c_int]
    L.sim_sync_distance.restype = C.c_long
    L.sim_sync_distance.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long]
    L.sim_rounds.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.sim_rounds_memo.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.sim_hyps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sim_rounds_keep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


KEEPS = [(4, 1), (4, 2), (4, 3), (4, 16), (8, 1), (8, 16)]      # (checkpoints per subsequence, trajectories kept)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    L = lib()
    port = Port()
    jpegs = synth.cfg3_imagenet_like(n, seed=3, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True)
    policies = [0, 1, 2, 3, 4, 5, 10]
    agg = {p: {"dist": [], "rounds_max": [], "rounds_sum": 0, "waves": 0, "passes": 0, "lanes": 0, "per_round": np.zeros(32, np.int64)} for p in policies}
    best6 = []
    memo = []
    keeps = {}
    hyps = {}
    for k, jp in enumerate(jpegs):
        pr = port.parse(jp)
        info = pr["info"]
        ecs = np.ascontiguousarray(pr["ecs"])
        nl = info["hsamp"] * info["vsamp"]
        dus = nl + (info["ncomp"] - 1)
        mcus = ((info["width"] + 8 * info["hsamp"] - 1) // (8 * info["hsamp"])) * ((info["height"] + 8 * info["vsamp"] - 1) // (8 * info["vsamp"]))
        bpm = len(ecs) / mcus
        dc_off = np.array(info["dc_offsets"], np.uint8); dc_sym = np.array(info["dc_symbols"], np.uint8)
        ac_off = np.array(info["ac_offsets"], np.uint8); ac_sym = np.array(info["ac_symbols"], np.uint8)
        dc_id = np.array(list(info["comp_dc"]), np.int32); ac_id = np.array(list(info["comp_ac"]), np.int32)
        s = L.sim_open(ecs.ctypes.data, len(ecs), dc_off.ctypes.data, dc_sym.ctypes.data, ac_off.ctypes.data, ac_sym.ctypes.data,
                       dc_id.ctypes.data, ac_id.ctypes.data, nl, dus)
        nlanes = (len(ecs) + S - 1) // S
        for p in policies:
            L.sim_policy(s, p)
            d = [L.sim_sync_distance(s, q * S * 8, 0, 64 * 1024 * 8) for q in range(1, nlanes)]
            agg[p]["dist"] += [(x, bpm) for x in d]
            out = np.zeros(40, np.int64)
            L.sim_rounds(s, S, out.ctypes.data)
            agg[p]["rounds_max"].append((int(out[0]), bpm))
            agg[p]["rounds_sum"] += int(out[1]); agg[p]["waves"] += int(out[2]); agg[p]["passes"] += int(out[3]); agg[p]["lanes"] += nlanes
            agg[p]["per_round"] += out[4:36]
        L.sim_policy(s, 0)
        out = np.zeros(40, np.int64)
        L.sim_rounds_memo(s, S, out.ctypes.data)
        memo.append((out.copy(), bpm, nlanes))
        for nchk in (4, 8):
            out = np.zeros(40, np.int64)
            L.sim_hyps(s, S, nchk, out.ctypes.data)
            hyps.setdefault(nchk, []).append((out.copy(), bpm))
        for key in KEEPS:
            out = np.zeros(40, np.int64)
            L.sim_rounds_keep(s, S, key[0], key[1], out.ctypes.data)
            keeps.setdefault(key, []).append((out.copy(), bpm))
        for q in range(1, nlanes):
            ds = [L.sim_sync_distance(s, q * S * 8, h, 64 * 1024 * 8) for h in range(dus)]
            ds = [x for x in ds if x >= 0]
            best6.append((min(ds) if ds else -1, bpm))
        L.sim_close(s)
    print(f"{n} pictures, S = {S} bytes")
    for p in policies:
        a = agg[p]
        d = np.array([x for x, _ in a["dist"]], float)
        dd = np.array([x for x, b in a["dist"] if b >= 150], float)
        ok = d[d >= 0] / 8
        okd = dd[dd >= 0] / 8
        rm = np.array([r for r, _ in a["rounds_max"]]); rmd = np.array([r for r, b in a["rounds_max"] if b >= 150])
        print(f"policy {p:2d}: sync distance mean {ok.mean():7.0f} B median {np.median(ok):6.0f} p90 {np.percentile(ok, 90):7.0f} | dense (>=150 B/MCU) mean {okd.mean():7.0f} median {np.median(okd):6.0f} p90 {np.percentile(okd, 90):7.0f}"
              f" | >S: {np.mean(ok > S):.2f} dense {np.mean(okd > S):.2f} | rounds per wave {a['rounds_sum'] / a['waves']:.2f}, slowest wave per picture mean {rm.mean():.2f} (dense {rmd.mean():.2f}) max {rm.max()} | re-sync passes per lane {a['passes'] / a['lanes']:.2f}"
              f" | active per round {list(a['per_round'][:10])}")
    b = np.array([x for x, _ in best6], float); bd = np.array([x for x, bb in best6 if bb >= 150], float)
    b = b[b >= 0] / 8; bd = bd[bd >= 0] / 8
    report_memo(memo)
    for nchk, v in hyps.items():
        for label, vv in (("all", v), ("dense", [m for m in v if m[1] >= 150])):
            t = np.sum([m[0] for m in vv], axis=0)
            print(f"hypotheses, {nchk} checkpoints [{label}]: links {t[0]}, without a merge {t[1]} ({t[1] / t[0]:.3f}), mean link {t[2] / t[0]:.0f} B; TRUE path: links {t[3]}, without a merge {t[4]} ({t[4] / max(t[3], 1):.3f}), mean {t[5] / max(t[3], 1):.0f} B, "
                  f"longest {max(m[0][7] for m in vv)} B; merged at checkpoint {list(t[9:9 + nchk])}; distinct exits per lane {t[20] / t[6]:.2f}")
    for key in KEEPS:
        v = keeps[key]
        tot = np.sum([m[0] for m in v], axis=0)
        dn = [m for m in v if m[1] >= 150]
        print(f"{key[0]} checkpoints, {key[1]:2d} trajectories kept: round time per wave {tot[0] / tot[2]:6.0f} B (rounds {tot[4] / tot[2]:.2f}); slowest wave per picture mean {np.mean([m[0][1] for m in v]):6.0f} B "
              f"(dense {np.mean([m[0][1] for m in dn]):6.0f}) max {max(m[0][1] for m in v)} B, rounds max {max(m[0][5] for m in v)}; bytes re-decoded per lane {tot[3] / sum(x[2] for x in memo):.0f}; merges into an older trajectory {tot[6]}")
    print(f"best of all start phases (policy 0): mean {b.mean():.0f} median {np.median(b):.0f} p90 {np.percentile(b, 90):.0f} | dense mean {bd.mean():.0f} median {np.median(bd):.0f} p90 {np.percentile(bd, 90):.0f} p99 {np.percentile(bd, 99):.0f}")



def report_memo(memo):
    tot = np.sum([m[0] for m in memo], axis=0)
    dense = [m for m in memo if m[1] >= 150]
    print(f"rounds with a memo per lane: decode rounds per wave {tot[1] / tot[2]:.2f} (all rounds {tot[4] / tot[2]:.2f}), slowest wave per picture: decode rounds mean {np.mean([m[0][0] for m in memo]):.2f} "
          f"(dense {np.mean([m[0][0] for m in dense]):.2f}) max {max(m[0][0] for m in memo)}, all rounds max {max(m[0][7] for m in memo)}; lane decodes per lane {tot[3] / sum(m[2] for m in memo):.2f}, memo hits {tot[5]}, most memo entries in a lane {max(m[0][6] for m in memo)}")


if __name__ == "__main__":
    main()

