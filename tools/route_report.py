"""GPU diagnostic: which decode route every golden fixture takes (parallel path / exact kernel up front / exact-kernel
fallback), its subsequence count and the duration of one decode.  Run on an MI355X box: python tools/route_report.py"""
import sys, os, json
sys.path.insert(0,'pim-jpeg-decoder_amd/python'); sys.path.insert(0,'tests')
import pjd_amd, numpy as np
M=json.load(open('tests/golden/manifest.json'))
ctx=pjd_amd.Context(0)
for n in sorted(M):
    if M[n]['rc']!=0: continue
    s=pjd_amd.Scanned(open(f'tests/golden/{n}.jpg','rb').read())
    with ctx.batch([s.desc]) as b:
        b.upload(); b.decode(); outs,st=b.download(); i=b.info()
        t,tot=b.decode_timed()
    print(f"{n:34s} seq={i['n_sequential']} fb={i['n_fallback']} subs={i['n_subsequences']:5d} st={st[0]} total_ms={tot:.3f}")
