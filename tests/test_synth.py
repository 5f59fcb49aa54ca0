"""The synthetic workload generator: its files are valid baseline JPEGs (Pillow decodes them close
to the source), and the oracle port agrees with the reference's own code on them."""
import hashlib
import io
import os
import sys

import numpy as np
import pytest

from conftest import golden_bytes, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255 ** 2 / max(mse, 1e-9))


@pytest.mark.parametrize("sub", [synth.SUB_444, synth.SUB_422, synth.SUB_420, synth.SUB_440, synth.SUB_GREY])
@pytest.mark.parametrize("ri", [0, 5])
def test_synth_files_are_valid_jpegs(sub, ri):
    PIL = pytest.importorskip("PIL.Image")
    rgb = synth.picture(150, 97, 11)
    data = synth.encode(rgb, 90, sub, ri)
    img = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
    ref = rgb if sub != synth.SUB_GREY else np.repeat((0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2])[..., None], 3, 2)
    assert img.shape == rgb.shape
    assert _psnr(img, ref) > 28


def test_synth_uses_annex_k_tables(port):
    """Same Huffman tables as libjpeg's defaults (a non-optimised Pillow file)."""
    a = port.parse(synth.make(64, 48, 1, 85, synth.SUB_420))["info"]
    b = port.parse(golden_bytes("env_64x48_420_q100"))["info"]
    for k in ("dc_offsets", "dc_symbols", "ac_offsets", "ac_symbols"):
        assert a[k][:2] == b[k][:2], k


def test_port_matches_ref_on_synth(port, ref, tmp_path):
    for i, (sub, ri, w, h) in enumerate([(synth.SUB_420, 0, 333, 200), (synth.SUB_444, 9, 120, 90), (synth.SUB_440, 0, 70, 130),
                                          (synth.SUB_422, 0, 200, 64), (synth.SUB_GREY, 3, 99, 99)]):
        data = synth.make(w, h, 100 + i, 85, sub, ri)
        jp = tmp_path / f"s{i}.jpg"
        jp.write_bytes(data)
        rc, out = ref.run_cli(str(jp), str(tmp_path / f"s{i}.bmp"))
        assert rc == 0 and out == ""
        o = port.decode(data)
        assert o["bmp"] == (tmp_path / f"s{i}.bmp").read_bytes()
    # the dense set's pictures: more detail, per-picture optimised Huffman tables
    for i, (w, h, q) in enumerate([(333, 200, 97), (96, 160, 88)]):
        data = synth.make(w, h, 200 + i, q, synth.SUB_420, 0, synth.DENSE_DETAIL, True)
        jp = tmp_path / f"d{i}.jpg"
        jp.write_bytes(data)
        rc, out = ref.run_cli(str(jp), str(tmp_path / f"d{i}.bmp"))
        assert rc == 0 and out == ""
        assert port.decode(data)["bmp"] == (tmp_path / f"d{i}.bmp").read_bytes()


def test_imagenet_like_specs_are_deterministic():
    a = synth.imagenet_like_specs(16)
    b = synth.imagenet_like_specs(16)
    assert a == b
    assert all(64 <= w <= 1024 and 64 <= h <= 1024 for w, h, _, _ in a)
    d1 = synth.make(*a[0][:3], a[0][3])
    d2 = synth.make(*a[0][:3], a[0][3])
    assert hashlib.sha256(d1).digest() == hashlib.sha256(d2).digest()


def test_dense_set_density_and_optimised_tables():
    """The default benchmark set: ImageNet-class density (>= 0.55 B/px) with four distinct Huffman tables per picture."""
    import pjd_amd
    import synth
    jp = synth.cfg3_imagenet_like(24, seed=3, detail=synth.DENSE_DETAIL, optimize=True, quality_shift=True)
    px = by = 0
    sets = set()
    for j in jp:
        s = pjd_amd.Scanned(j)
        assert s.valid
        d = s.desc
        px += int(d.width) * int(d.height)
        by += int(d.ecs_len)
        sets.add(bytes(d.dc[0].symbols[:12]) + bytes(d.ac[0].offsets) + bytes(d.ac[1].offsets))
    assert by / px >= 0.55, by / px
    assert len(sets) >= 20          # per-picture tables, not one shared set
