"""CPU tests that pin the oracle (no GPU).

* oracle/_ref (the reference's own scanner / Huffman / BMP code + dpu_stages.c)
  against the SURVEY section 0.4 known-answer hash;
* oracle/liboracle.so (plain-C port) against oracle/_ref, field by field;
* both against the committed manifest (generated from oracle/_ref).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import golden_bytes

SURVEY_SHA256 = "11ab0c81cfc918410245c5ff0923f787219521073c094cbfd7e763f4b3444c1f"
SURVEY_MD5 = "fa708c3f78f341909052666db44586d2"
SURVEY_HEAD = "424d5e950800000000001a0000000c000000f4017701010018009ba59e99a39c"

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(json.load(open(os.path.join(HERE, "golden", "manifest.json"))).keys())


def test_known_answer_port(port):
    out = port.decode(golden_bytes("ilsvrc_val_00000001"))
    bmp = out["bmp"]
    assert len(bmp) == 562526
    assert bmp[:32].hex() == SURVEY_HEAD
    assert hashlib.md5(bmp).hexdigest() == SURVEY_MD5
    assert hashlib.sha256(bmp).hexdigest() == SURVEY_SHA256


def test_known_answer_ref(ref, tmp_path):
    jp = tmp_path / "a.jpg"
    jp.write_bytes(golden_bytes("ilsvrc_val_00000001"))
    rc, _ = ref.run_cli(str(jp), str(tmp_path / "a.bmp"))
    assert rc == 0
    assert hashlib.sha256((tmp_path / "a.bmp").read_bytes()).hexdigest() == SURVEY_SHA256


@pytest.mark.parametrize("name", NAMES)
def test_port_matches_manifest(port, manifest, name):
    ent = manifest[name]
    out = port.decode(golden_bytes(name), name="{path}")
    if ent["rc"] != 0:
        assert not out["valid"]
        assert out["log"] == ent["stdout"]
        return
    assert out["valid"]
    assert out["log"] == ent["stdout"]
    assert (out["huff_rc"] == 0) == bool(ent["huff_ok"])
    assert hashlib.sha256(out["coef"].tobytes()).hexdigest() == ent["coef_sha256"]
    assert len(out["bmp"]) == ent["bmp_len"]
    assert hashlib.sha256(out["bmp"]).hexdigest() == ent["bmp_sha256"]
    i = out["info"]
    assert [i["width"], i["height"], i["ncomp"], i["hsamp"], i["vsamp"], i["restart_interval"]] == ent["dims"]


@pytest.mark.parametrize("name", NAMES)
def test_port_matches_ref_fields(port, ref, tmp_path, name):
    data = golden_bytes(name)
    jp = tmp_path / (name + ".jpg")
    jp.write_bytes(data)
    r = ref.parse_and_huffman(str(jp))
    p = port.parse(data)
    assert r is not None
    assert r["info"]["valid"] == p["info"]["valid"]
    if not r["info"]["valid"]:
        return
    for k, v in r["info"].items():
        assert p["info"][k] == v, k
    assert np.array_equal(r["ecs"], p["ecs"])
    assert np.array_equal(r["metadata"], p["metadata"])


def test_stage_entry_points_compose(port):
    """dequant -> idct -> colour one by one == orc_dpu_exec."""
    out = port.decode(golden_bytes("env_61x45_420_q100_opt"))
    meta = out["metadata"]
    a = out["coef"][0].copy()
    b = out["coef"][0].copy()
    port.dpu_exec(meta, a)
    for s in range(3):
        port.dpu_stage(meta, b, s)
    assert np.array_equal(a, b)


WRAP = [n for n in NAMES if n.startswith("wrap_")]


@pytest.mark.parametrize("name", WRAP)
def test_wrap_fixtures_do_wrap(port, name):
    """The wrap_* fixtures exist to push int16 truncation and 32-bit product overflow through the whole path (reference
    src/decoder_dpu.c:169-172 dequantise, :218-320 IDCT stores, :376-382 colour products).  Check on the oracle's own stage
    outputs that each one does what it claims: products beyond int16 in the dequantiser, IDCT outputs using the full int16
    range, and (colour pictures) chroma samples far outside +-365 / +-288, where 5880414 * Cr and 7432306 * Cb leave 32 bits."""
    o = port.decode(golden_bytes(name))
    assert o["valid"] and o["huff_rc"] == 0
    meta, coef = o["metadata"], o["coef"].astype(np.int64)
    ncomp = int(meta[4])
    wrapped = 0
    blocks = coef.reshape(-1, 3, 4, 64)                       # [blk16][component][position][natural index]
    for c in range(ncomp):
        q = meta[20 + 64 * int(meta[7 + c]):][:64].astype(np.int64)
        prod = blocks[:, c] * q
        wrapped += int((np.abs(prod) > 32767).sum())
    assert wrapped > 300, wrapped
    if name.endswith("q65535"):
        return                    # coef * 65535 = -coef modulo 2^16: the truncation fires on every product, the values stay small
    st = o["coef"].copy()
    for d in range(st.shape[0]):
        port.dpu_stage(meta, st[d], 0)
        port.dpu_stage(meta, st[d], 1)
    after_idct = st.reshape(-1, 3, 4, 64)
    assert int(np.abs(after_idct[:, 0].astype(np.int32)).max()) > 20000            # int16 stores of the IDCT passes are exercised
    if ncomp == 3:
        assert int((np.abs(after_idct[:, 2].astype(np.int32)) > 365).sum()) > 100   # Cr: 5880414 * Cr overflows int32
        assert int((np.abs(after_idct[:, 1].astype(np.int32)) > 288).sum()) > 100   # Cb: 7432306 * Cb overflows int32
