"""which of a few synthetic pictures the parallel decoder flags, and why (flag_waves by reason)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.dirname(__file__))
import pjd_amd, synth
D = synth.DENSE_DETAIL
pics = {"420 q97 1536x1024": synth.make(1536, 1024, 11, quality=97, subsampling=synth.SUB_420, detail=D, optimize=True),
        "444 q95 rst7": synth.make(640, 480, 12, quality=95, subsampling=synth.SUB_444, restart_interval=7, detail=D, optimize=True),
        "grey q96": synth.make(800, 600, 13, quality=96, subsampling=synth.SUB_GREY, detail=D, optimize=True),
        "444 q95 rst7 plain tables": synth.make(640, 480, 12, quality=95, subsampling=synth.SUB_444, restart_interval=7, detail=D, optimize=False),
        "444 q95 no rst": synth.make(640, 480, 12, quality=95, subsampling=synth.SUB_444, detail=D, optimize=True)}
c = pjd_amd.Context(0)
for name, b in pics.items():
    s = pjd_amd.Scanned(b)
    with c.batch([s.desc]) as bt:
        bt.upload(); bt.decode(); bt.sync()
        i = bt.info()
        print(name, len(b), "bytes; seq", i["n_sequential"], "fallback", i["n_fallback"], "flags", i["flag_waves"], "waves", i["n_huff_waves"], "sub", i["sub_bytes"], "segments", int(s.desc.n_segments), "walks", i["walks"], i["walk_lanes"], "rounds", i["sync_rounds"], i["fix_rounds"])
