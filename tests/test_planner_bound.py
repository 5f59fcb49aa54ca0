"""The bound behind the size of the lane streams (pjd_plan.cpp, min_step_bits_x256), recomputed another way.

A lane's region holds 8 * S / mu + slack step words, mu = the fewest bits of stream per write-pass step that any stream coded with the
picture's Huffman tables can sustain: the minimum mean weight of a cycle in the step graph (node D: a unit's DC symbol comes next;
node A_L: an AC symbol with a code of >= L bits comes next; edges: the single steps and the pairs the decode tables hold, weighted by
the bits they consume).  The planner finds it with Karp's algorithm over edge minima gathered in one pass over the symbols; here the
graph is built naively from the same rules (include/pjd.h, pjd_internal.h: what a pair is) and the minimum mean cycle comes from a
bisection on lambda with Bellman-Ford negative-cycle detection.  Too small a mu wastes HBM; too large a one would send pictures to the
exact kernel through the overflow flag -- never a wrong picture, but a 1000x slower one."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

INF = 1 << 30


def _symbols(t, ac):
    out = []
    for ln in range(1, 17):
        for q in range(t.offsets[ln - 1], min(t.offsets[ln], 162)):
            s = t.symbols[q]
            valid = s != 0xFF and ((s & 15) <= 10 if ac else s <= 11)      # the reference's "no symbol" / out-of-range size: code bits only, never paired
            out.append((ln, ln + (((s & 15) if ac else s) if valid else 0), valid and ac and s == 0, valid))
    return out


def _graph(d):
    V = 17
    w = np.full((V, V), INF, dtype=np.int64)

    def edge(u, v, bits):
        w[u, v] = min(w[u, v], bits)

    need = lambda bits: 1 if bits >= 9 else 10 - bits                      # the code after a symbol that stayed single is at least this long
    combos = [(int(d.comp_dc[c]), int(d.comp_ac[c])) for c in range(d.num_components)]
    for dcs, acs in set(combos):                                          # every component's DC table pairs with its own AC table (the planner
        dc, ac = _symbols(d.dc[dcs], False), _symbols(d.ac[acs], True)    # gives a DC table one slot per AC table it is used with)
        pairs = True
        for (l, b, e, val) in dc:
            edge(0, need(b) if (pairs and val) else 1, b)
            if pairs and val and b <= 8:
                for (l2, b2, e2, v2) in ac:
                    if v2 and b + l2 <= 9:
                        edge(0, 0, b + b2)                                 # the unit may end with the second symbol (EOB, or slot 63 filled)
                        if not e2:
                            edge(0, 1, b + b2)
        for L in range(1, 17):
            for (l, b, e, val) in ac:
                if l < L:
                    continue
                edge(L, 0, b)
                if e:
                    continue
                edge(L, need(b) if val else 1, b)
                if val and b <= 8:
                    for (l2, b2, e2, v2) in ac:
                        if v2 and b + l2 <= 9:
                            edge(L, 0, b + b2)
                            if not e2:
                                edge(L, 1, b + b2)
    return w


def _has_cycle_below(w, lam):
    """Is there a cycle of mean weight < lam?  Bellman-Ford on w - lam from a virtual source."""
    V = w.shape[0]
    dist = np.zeros(V)
    for _ in range(V):
        changed = False
        for u in range(V):
            for v in range(V):
                if w[u, v] < INF and dist[u] + w[u, v] - lam < dist[v] - 1e-12:
                    dist[v] = dist[u] + w[u, v] - lam
                    changed = True
        if not changed:
            return False
    return True


def _min_mean_cycle(w):
    lo, hi = 0.0, 64.0
    for _ in range(40):
        mid = (lo + hi) / 2
        if _has_cycle_below(w, mid):
            hi = mid
        else:
            lo = mid
    return (lo + hi) / 2


def _cases():
    import synth
    yield "annex-K 4:2:0", synth.make(320, 240, 1, 85, synth.SUB_420, 0)
    yield "annex-K 4:4:4 + DRI", synth.make(200, 160, 2, 90, synth.SUB_444, 5)
    yield "annex-K grey", synth.make(160, 120, 3, 75, synth.SUB_GREY, 0)
    yield "fitted dense 4:2:0", synth.make(500, 375, 4, 95, synth.SUB_420, 0, synth.DENSE_DETAIL, True)
    yield "fitted dense 4:2:2", synth.make(400, 300, 5, 90, synth.SUB_422, 0, synth.DENSE_DETAIL, True)
    yield "fitted flat q5 4:4:4", synth.make(649, 513, 92, 5, synth.SUB_444, 0, 1.0, True)
    yield "fitted flat q5 grey", synth.make(300, 200, 94, 5, synth.SUB_GREY, 0, 1.0, True)
    yield "fitted q30 4:4:0", synth.make(264, 400, 6, 30, synth.SUB_440, 0, 1.0, True)


def test_step_bound_equals_an_independent_minimum_mean_cycle():
    import pjd_amd
    seen = []
    for label, jpeg in _cases():
        s = pjd_amd.Scanned(jpeg)
        assert s.valid, label
        got = pjd_amd.plan_step_bits(s.desc)
        want = _min_mean_cycle(_graph(s.desc))
        assert got <= want + 1e-6 and want - got < 1.0 / 256 + 1e-6, (label, got, want)      # the planner rounds down to 1/256
        seen.append(got)
    assert min(seen) <= 2.0 and max(seen) >= 4.0            # the flat pictures' one-bit codes and the dense pictures' tables are both in the set


def test_known_cycle_of_the_annex_k_tables():
    """Annex-K tables, colour: a chroma unit of DC difference 0 ('00'), one +-1 coefficient at slot 1 ('01' + its sign bit) and an EOB
    ('00') is two steps -- the pair DC + (0,1), then the EOB alone -- of 7 bits: 3.5 bits per step, and nothing is cheaper."""
    import pjd_amd
    import synth
    s = pjd_amd.Scanned(synth.make(64, 48, 7, 85, synth.SUB_420, 0))
    assert pjd_amd.plan_step_bits(s.desc) == 3.5
