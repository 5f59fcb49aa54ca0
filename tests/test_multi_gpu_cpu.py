"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the sharding helpers
(image LPT split, descriptor broadcast, ECS scatter, per-rank shard descriptors and their plans).
No GPU compute here; the GPU side of sharding is covered by test_gpu_parity.py::test_sharded_*."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import golden_bytes, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tall_picture():
    """64 x 16384, 4:4:4, one restart interval per MCU row: 2,048 restart segments like BASELINE config 5's 16384 x 16384 picture
    (whose descriptor has the same shape: 2,048 offsets), at 1/256 of its pixels."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    return synth.make(64, 16384, 5, 85, synth.SUB_444, 8)


def _worker(rank, world, port, ret, which="fixture"):
    sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import pjd_amd
    from pjd_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the code path of bench.py --workload cfg5split: rank 0 scans, one descriptor broadcast, bitstream slices scattered
        data = (golden_bytes("rstrow_200x150_444_opt") if which == "fixture" else _tall_picture()) if rank == 0 or which == "fixture" else None
        s = pjd_amd.Scanned(data) if rank == 0 else None
        d, keep, blob_bytes = parallel.distribute_image(s, src=0)
        sl, segs = keep
        desc = d
        info = pjd_amd.plan_info([d])
        out = {"w": int(desc.width), "h": int(desc.height), "nseg": int(desc.n_segments), "first": int(d.shard_first_seg),
                     "count": int(d.shard_n_segs), "slice": len(sl), "subs": info["n_subsequences"], "ri": int(desc.restart_interval),
                     "seq": info["n_sequential"], "sha": __import__("hashlib").sha256(sl.tobytes()).hexdigest(), "blob": int(blob_bytes),
                     "own_first_off": int(segs[d.shard_first_seg]), "own_last_off": int(segs[d.shard_first_seg + d.shard_n_segs - 1]),
                     "world": dist.get_world_size()}
        if which == "fixture":
            out["lo"] = int(pjd_amd.Scanned(data).seg_offsets()[d.shard_first_seg])
        ret[rank] = out
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_descriptor_broadcast_and_scatter():
    import hashlib
    import pjd_amd
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    s = pjd_amd.Scanned(golden_bytes("rstrow_200x150_444_opt"))
    segs, ecs = s.seg_offsets(), s.ecs()
    assert ret[0]["nseg"] == ret[1]["nseg"] == len(segs) == 19
    assert (ret[0]["w"], ret[0]["h"]) == (ret[1]["w"], ret[1]["h"]) == (200, 150)
    assert ret[0]["first"] == 0 and ret[0]["count"] == 10 and ret[1]["first"] == 10 and ret[1]["count"] == 9
    cut = int(segs[10])
    assert ret[0]["slice"] == cut and ret[1]["slice"] == len(ecs) - cut and ret[1]["lo"] == cut
    assert ret[0]["sha"] == hashlib.sha256(ecs[:cut].tobytes()).hexdigest()
    assert ret[1]["sha"] == hashlib.sha256(ecs[cut:].tobytes()).hexdigest()
    assert ret[0]["seq"] == 0 and ret[1]["seq"] == 0 and ret[0]["subs"] > 0 and ret[1]["subs"] > 0


def test_eight_ranks_tile_a_2048_segment_picture_once():
    """World size 8 (gloo) through parallel.distribute_image -- what `bench.py --gpus 8 --workload cfg5split` runs over RCCL -- on a
    picture with 2,048 restart segments (BASELINE config 5's count): every rank gets 256 consecutive segments, the slices are the
    scanned bitstream cut at segment boundaries (each byte exactly once, in order), every shard's own offsets start at 0, and the
    MCU ranges they stand for tile the picture once.  Reference: one picture over every allocated DPU, src/decoder_host.cpp:125-149,262-312."""
    import hashlib
    import pjd_amd
    world, port = 8, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, "tall"), nprocs=world, join=True)
    s = pjd_amd.Scanned(_tall_picture())
    segs, ecs = s.seg_offsets(), s.ecs()
    assert len(segs) == 2048 and s.desc.restart_interval == 8
    n_mcu = 8 * 2048
    next_seg, next_byte, next_mcu = 0, 0, 0
    for r in range(world):
        g = ret[r]
        assert g["world"] == 8 and (g["w"], g["h"], g["nseg"], g["ri"]) == (64, 16384, 2048, 8)
        assert (g["first"], g["count"]) == (next_seg, 256)
        lo = int(segs[g["first"]])
        hi = int(segs[g["first"] + g["count"]]) if g["first"] + g["count"] < 2048 else len(ecs)
        assert lo == next_byte and g["slice"] == hi - lo
        assert g["sha"] == hashlib.sha256(ecs[lo:hi].tobytes()).hexdigest()
        assert g["own_first_off"] == 0 and g["own_last_off"] == int(segs[g["first"] + g["count"] - 1]) - lo
        m0, m1 = g["first"] * 8, min((g["first"] + g["count"]) * 8, n_mcu)
        assert m0 == next_mcu
        assert g["seq"] == 0 and g["subs"] >= 256          # at least a lane per restart segment, nothing routed to the exact kernel
        assert g["blob"] == ret[0]["blob"] > 2048 * 8
        next_seg, next_byte, next_mcu = g["first"] + g["count"], hi, m1
    assert (next_seg, next_byte, next_mcu) == (2048, len(ecs), n_mcu)


def test_more_ranks_than_segments_leaves_ranks_idle():
    """A rank without restart segments gets no descriptor (shard_n_segs == 0 would mean "decode everything")."""
    import pjd_amd
    from pjd_amd import parallel
    s = pjd_amd.Scanned(golden_bytes("rst7_gray_61x45"))
    segs, ecs = s.seg_offsets(), s.ecs()
    world = len(segs) + 3
    got = 0
    for r in range(world):
        f, c = parallel.segment_range(len(segs), r, world)
        lo = int(segs[f]) if c else 0
        hi = (int(segs[f + c]) if f + c < len(segs) else len(ecs)) if c else 0
        d, keep = parallel.shard_descriptor(s.desc, segs, ecs[lo:hi], lo, r, world)
        if c == 0:
            assert d is None
        else:
            assert d.shard_n_segs == c and pjd_amd.plan_info([d])["n_sequential"] == 0
            got += c
    assert got == len(segs)


def test_lpt_and_segment_ranges():
    from pjd_amd import parallel
    costs = [9, 1, 8, 2, 7, 3, 6, 4, 5]
    parts = parallel.lpt_assign(costs, 3)
    assert sorted(sum(parts, [])) == list(range(9))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 2
    for n in (1, 7, 8, 19, 2048):
        for w in (1, 2, 4, 8):
            got = [parallel.segment_range(n, r, w) for r in range(w)]
            assert sum(c for _, c in got) == n
            assert all(got[r][0] + got[r][1] == got[r + 1][0] for r in range(w - 1))


# ---- the C++ split path (include/pjd.h: pjd_split_plan / pjd_split_decode): range and offset arithmetic, no device ------------
@pytest.mark.parametrize("name", ["rstrow_200x150_444_opt", "rst4_128x96_444", "rstrow_gray_100x60", "rst1_61x45_444", "rst7_gray_61x45"])
@pytest.mark.parametrize("world", [1, 2, 3, 5, 8, 16])
def test_split_plan_matches_the_python_sharding(name, world):
    """pjd_split_plan (what pjd_split_decode and `bin/decoder --split` use) against pjd_amd.parallel (what bench.py's
    cfg5split uses): same segment ranges, same byte ranges, same rebased offsets; the ranges tile the picture's segments,
    bytes and MCUs exactly once; ranks beyond the number of segments get nothing."""
    import pjd_amd
    from pjd_amd import parallel
    s = pjd_amd.Scanned(golden_bytes(name))
    d = s.desc
    segs, ecs = s.seg_offsets(), s.ecs()
    nseg = len(segs)
    hs, vs = int(d.h_samp), int(d.v_samp)
    n_mcu = (((d.width + 7) // 8 + hs - 1) // hs) * (((d.height + 7) // 8 + vs - 1) // vs)
    next_seg, next_byte, next_mcu = 0, 0, 0
    for r in range(world):
        f, c = parallel.segment_range(nseg, r, world)
        got = pjd_amd.split_plan(d, world, r)
        if c == 0:
            assert got is None
            continue
        assert (got["first_seg"], got["n_segs"]) == (f, c)
        lo = int(segs[f]); hi = int(segs[f + c]) if f + c < nseg else len(ecs)
        assert (got["byte_lo"], got["byte_hi"], got["ecs_len"], got["ecs_delta"]) == (lo, hi, hi - lo, lo)
        sd, keep = parallel.shard_descriptor(d, segs, ecs[lo:hi], lo, r, world)
        assert np.array_equal(got["seg_offsets"], keep[1])
        assert got["first_mcu"] == min(f * d.restart_interval, n_mcu) and got["last_mcu"] == min((f + c) * d.restart_interval, n_mcu)
        assert (f, lo, got["first_mcu"]) == (next_seg, next_byte, next_mcu)
        next_seg, next_byte, next_mcu = f + c, hi, got["last_mcu"]
    assert (next_seg, next_byte, next_mcu) == (nseg, len(ecs), n_mcu)


def test_split_plan_rejects_pictures_without_restart_segments():
    import pjd_amd
    s = pjd_amd.Scanned(golden_bytes("big_640x480_420_q85"))
    with pytest.raises(pjd_amd.PjdError):
        pjd_amd.split_plan(s.desc, 2, 0)
