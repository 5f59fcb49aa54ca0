"""CPU tests of the product's host scanner / BMP helper (libpjdhost.so) against the oracle,
and of the C-ABI libraries' export tables.  No GPU needed, no compute calls."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import golden_bytes, ROOT

import pjd_amd

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(json.load(open(os.path.join(HERE, "golden", "manifest.json"))).keys())


@pytest.mark.parametrize("name", NAMES)
def test_scanner_matches_oracle(port, manifest, name):
    data = golden_bytes(name)
    s = pjd_amd.Scanned(data, name="{path}")
    o = port.parse(data, name="{path}")
    ent = manifest[name]
    assert s.valid == bool(o["info"]["valid"]) == (ent["rc"] == 0)
    if not s.valid:
        # rejected files: same messages as the reference (the manifest's stdout comes from it)
        assert s.log == ent["stdout"]
        return
    assert s.log == ""
    d, i = s.desc, o["info"]
    assert (d.width, d.height, d.num_components, d.h_samp, d.v_samp) == (i["width"], i["height"], i["ncomp"], i["hsamp"], i["vsamp"])
    assert d.restart_interval == i["restart_interval"]
    n = d.num_components
    for f, g in (("comp_h", "comp_h"), ("comp_v", "comp_v"), ("comp_qt", "comp_qt"), ("comp_dc", "comp_dc"), ("comp_ac", "comp_ac")):
        assert list(getattr(d, f))[:n] == i[g][:n], f
    assert list(d.qt_set) == i["qt_set"]
    for t in range(4):
        if i["qt_set"][t]:
            assert list(d.qt[t]) == i["qt"][t]
        assert d.dc[t].set == i["dc_set"][t] and d.ac[t].set == i["ac_set"][t]
        if i["dc_set"][t]:
            assert list(d.dc[t].offsets) == i["dc_offsets"][t]
            k = i["dc_offsets"][t][16]
            assert list(d.dc[t].symbols)[:k] == i["dc_symbols"][t][:k]
        if i["ac_set"][t]:
            assert list(d.ac[t].offsets) == i["ac_offsets"][t]
            k = i["ac_offsets"][t][16]
            assert list(d.ac[t].symbols)[:k] == i["ac_symbols"][t][:k]
    assert np.array_equal(s.ecs(), o["ecs"])
    assert np.array_equal(s.metadata(), o["metadata"])
    segs = s.seg_offsets()
    assert segs[0] == 0 and np.all(np.diff(segs.astype(np.int64)) >= 0) and segs[-1] <= d.ecs_len


def test_segment_offsets_are_restart_boundaries(port):
    """Each recorded offset is where the reference's BitReader stands after align()."""
    for name in ("rst4_128x96_444", "rstrow_200x150_444_opt", "rst7_gray_61x45", "rst1_61x45_444"):
        data = golden_bytes(name)
        s = pjd_amd.Scanned(data)
        d = s.desc
        mcux = (d.width + 8 * d.h_samp - 1) // (8 * d.h_samp)
        mcuy = (d.height + 8 * d.v_samp - 1) // (8 * d.v_samp)
        assert d.n_segments == -(-mcux * mcuy // d.restart_interval)
        # raw file: count RSTn markers in the scan
        raw = np.frombuffer(data, np.uint8)
        n_rst = sum(1 for i in range(len(raw) - 1) if raw[i] == 0xFF and 0xD0 <= raw[i + 1] <= 0xD7)
        assert d.n_segments == n_rst + 1


def test_rgb_to_bmp_matches_oracle(port):
    for name in ("env_61x45_444_q30_opt", "env_17x9_420_q100", "ilsvrc_val_00000001"):
        if name not in NAMES:
            continue
        o = port.decode(golden_bytes(name))
        assert pjd_amd.rgb_to_bmp(o["rgb"]) == o["bmp"]


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pjd_[a-z0-9_]+)\s*\(", txt)))


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_libpjd_exports_every_declared_symbol():
    names = _declared("pjd.h")
    assert "pjd_batch_decode" in names and "pjd_exec_dpu_payload" in names
    exp = _exported(pjd_amd.LIBPJD)
    missing = [n for n in names if n not in exp]
    assert not missing, missing
    L = pjd_amd.dev_lib()          # loads (no compute call)
    assert L.pjd_version() == pjd_amd.ABI_VERSION
    assert L.pjd_output_size(500, 375, pjd_amd.OUT_BMP) == 562526
    assert L.pjd_output_size(61, 45, pjd_amd.OUT_BMP) == 26 + 45 * (183 + 1)


def test_libpjdhost_exports_every_declared_symbol():
    names = [n for n in _declared("pjd_host.h") if n not in _declared("pjd.h")]
    exp = _exported(pjd_amd.LIBHOST)
    missing = [n for n in names if n not in exp]
    assert not missing, missing


def test_libpjdpipe_exports_every_declared_symbol():
    names = [n for n in _declared("pjd_pipeline.h") if n not in _declared("pjd.h") and n not in ("pjd_pipe_sink",)]
    assert "pjd_pipe_run_files" in names and "pjd_pipe_run_memory" in names
    exp = _exported(pjd_amd.LIBPIPE)
    missing = [n for n in names if n not in exp]
    assert not missing, missing
    pjd_amd.pipe_lib()             # loads together with libpjd / libpjdhost (no compute call)


def test_integration_option_a_snippet_compiles(tmp_path):
    """INTEGRATION.md option A (the literal replacement for the reference's DPU dispatch, src/decoder_host.cpp:268-312):
    the code block is taken from the document as is and compiled in a stand-alone translation unit that declares
    nothing but the two vectors of the reference's `Batch` the snippet touches (decoder_host.cpp:24-29)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## Option A"):]
    code = sec[sec.index("```cpp") + 6:]
    code = code[:code.index("```")]
    head, body = code[:code.index("// in offloading()")], code[code.index("// in offloading()"):]
    src = tmp_path / "option_a.cpp"
    src.write_text("""#include <cstdint>
#include <cstring>
#include <exception>
#include <iostream>
#include <vector>
typedef unsigned int uint;
struct Batch { std::vector<std::vector<short>> mcus; std::vector<std::vector<uint32_t>> metadata; };
""" + head + "\nint offloading_dispatch(Batch &batch)\n{\n" + body + "\nreturn rc;\n}\n")
    p = subprocess.run(["g++", "--std=c++11", "-Wall", "-Werror", "-Wno-unused-variable", "-c", str(src), "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "option_a.o")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # and it links against the library it names
    p = subprocess.run(["g++", "-shared", "-fPIC", "--std=c++11", str(src), "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(pjd_amd.LIBPJD),
                        "-lpjd", "-Wl,-rpath," + os.path.dirname(pjd_amd.LIBPJD), "-o", str(tmp_path / "option_a.so")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_pipe_assign_deals_longest_first_to_least_loaded():
    """The multi-device batcher's dealing rule (pjd_pipe_assign, no GPU): LPT on input bytes, deterministic,
    within 4/3 of the best possible makespan (Graham's bound), every device used when there are enough batches."""
    import itertools
    import numpy as np
    assert pjd_amd.pipe_assign([], 3) == []
    assert pjd_amd.pipe_assign([5, 5, 5], 1) == [0, 0, 0]
    assert pjd_amd.pipe_assign([7, 1, 1, 1, 1, 1, 1, 1], 2) == [0, 1, 1, 1, 1, 1, 1, 1]
    assert pjd_amd.pipe_assign([3, 3, 3, 3], 4) == [0, 1, 2, 3]          # ties: lower index, lower device
    assert pjd_amd.pipe_assign([0, 0, 0, 0], 2) == [0, 1, 0, 1]          # empty batches still alternate
    # size-sorted inputs in consecutive batches (what the CLI produces): ascending costs
    rng = np.random.default_rng(5)
    for n_dev, n in ((2, 9), (4, 10), (8, 40), (3, 7)):
        cost = np.sort(rng.integers(1, 10 ** 9, n)).tolist()
        dev = pjd_amd.pipe_assign(cost, n_dev)
        assert dev == pjd_amd.pipe_assign(cost, n_dev)
        load = [sum(c for c, d in zip(cost, dev) if d == k) for k in range(n_dev)]
        assert all(load), load
        if n <= 10:        # brute force the optimum
            best = min(max(sum(c for c, d in zip(cost, a) if d == k) for k in range(n_dev)) for a in itertools.product(range(n_dev), repeat=n))
            assert max(load) <= best * (4 / 3 - 1 / (3 * n_dev)) + 1
        else:
            assert max(load) <= max(sum(cost) / n_dev, max(cost)) * 4 / 3
    with pytest.raises(pjd_amd.PjdError):
        pjd_amd.pipe_assign([1], 0)


def test_plan_routes_odd_huffman_tables_to_exact_kernel():
    """Planner only (no GPU): tables with more long-code prefixes than the LDS budget, or not a prefix code at all."""
    for name in ("huff_longtail_96x64_444", "huff_oversub_96x64_444"):
        s = pjd_amd.Scanned(golden_bytes(name))
        assert s.valid
        assert pjd_amd.plan_info([s.desc])["n_sequential"] == 1, name
    s = pjd_amd.Scanned(golden_bytes("big_500x375_444_q92_opt"))      # optimised tables with a normal tail: parallel path
    assert pjd_amd.plan_info([s.desc])["n_sequential"] == 0


def test_plan_info_host_only():
    """The planner runs without a device: lanes, data units, routing."""
    descs, keep = [], []
    for name in ("ilsvrc_val_00000001", "big_640x480_420_q85", "rstrow_200x150_444_opt", "div_rst_420_64x48", "gray_61x45"):
        s = pjd_amd.Scanned(golden_bytes(name))
        keep.append(s)
        descs.append(s.desc)
    info = pjd_amd.plan_info(descs)
    assert info["n_images"] == 5
    assert info["pixels"] == 500 * 375 + 640 * 480 + 200 * 150 + 64 * 48 + 61 * 45
    # 4:4:4 -> 3 units per 8x8; 4:2:0 -> 6 per 16x16; grey -> 1 per 8x8
    assert info["n_data_units"] == 63 * 47 * 3 + 40 * 30 * 6 + 25 * 19 * 3 + 4 * 3 * 6 + 8 * 6
    assert info["n_sequential"] == 1          # 4:2:0 + DRI under the reference's restart rule
    assert info["n_subsequences"] >= (info["ecs_bytes"] - 3000) // 128


def test_progressive_files_are_rejected_by_default_and_parsed_on_request():
    """The reference rejects a progressive file at its first inter-scan marker (src/jpeg_scanner.cpp:425-430) and so does the
    scanner by default -- same message, fixture neg_progressive_64x48.  With PJD_SCAN_PROGRESSIVE (not reference
    behaviour) the same file yields its scan list: a DC-first scan over all components first, spectral selections inside 1..63,
    successive approximation consistent from scan to scan, tables attached to every scan that reads symbols; the planner routes
    the picture to the scan-by-scan kernel (one dense-scratch picture, no Huffman lanes)."""
    data = golden_bytes("neg_progressive_64x48")
    s0 = pjd_amd.Scanned(data)
    assert not s0.valid and "Invalid marker during compressed data scan: 0xc4" in s0.log
    s = pjd_amd.Scanned(data, options=pjd_amd.SCAN_PROGRESSIVE)
    assert s.valid and s.log == "" and int(s.desc.flags) & pjd_amd.F_PROGRESSIVE
    n = int(s.desc.n_scans)
    assert n >= 4 and int(s.desc.ecs_len) == 0
    first = s.desc.scans[0]
    assert (first.n_comp, first.ss, first.se, first.ah) == (3, 0, 0, 0)
    al_seen = {}
    for k in range(n):
        sc = s.desc.scans[k]
        comps = tuple(sc.comp[q] for q in range(sc.n_comp))
        assert 1 <= sc.n_comp <= 3 and sc.ss <= sc.se <= 63 and (sc.ss == 0) == (sc.se == 0)
        assert sc.ss == 0 or sc.n_comp == 1
        for c in comps:
            for z in range(sc.ss, sc.se + 1):
                prev = al_seen.get((c, z))
                assert (sc.ah == 0 and prev is None) or (prev is not None and sc.ah == prev and sc.al == prev - 1), (k, c, z)
                al_seen[(c, z)] = sc.al
        if not (sc.ss == 0 and sc.ah != 0):
            assert all(sc.table[q].set for q in range(sc.n_comp))
        assert sc.ecs_len > 0
    assert all(al_seen[(c, z)] == 0 for c in range(3) for z in range(64))      # every coefficient refined down to bit 0
    info = pjd_amd.plan_info([s.desc])
    assert info["n_sequential"] == 1 and info["n_subsequences"] == 0 and info["n_data_units"] == 4 * 3 * 6       # 64x48, 4:2:0
    # a baseline file is the same with and without the option
    b = golden_bytes("env_64x48_444_q85")
    a1, a2 = pjd_amd.Scanned(b), pjd_amd.Scanned(b, options=pjd_amd.SCAN_PROGRESSIVE)
    assert a1.valid and a2.valid and int(a2.desc.flags) == 0 and bytes(a1.ecs()) == bytes(a2.ecs())
