// sync_sim.c -- CPU model of the parallel entropy decoder's self-synchronisation (measurement tooling, not product code).
//
// Question it answers: how far does a speculative decoder started at a subsequence boundary run before it is in step with the
// true decoder, and how many re-sync rounds does a wave of 64 subsequences need, under different SPECULATION POLICIES
// (what a state-only pass does when it meets something a valid stream never holds: a run past slot 63, an unassigned code).
// The true decode (reference src/jpeg_scanner.cpp:467-520) is not affected by the policy: a valid stream never triggers it.
//
// build: gcc -O2 -fPIC -shared -o tools/libsyncsim.so tools/sync_sim.c
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint8_t len; uint8_t sym; } Ent;      // len 0: no code matches these 16 bits

typedef struct {
    const uint8_t *ecs; long nbits;
    Ent *dc[3], *ac[3];
    int nluma, dus;
    uint8_t *tz, *tc;          // true trajectory: tz[p] = z + 1 where a symbol starts at bit p (0: none), tc[p] = unit phase
    int policy;
} Sim;

static Ent *build(const uint8_t *offsets, const uint8_t *symbols)
{
    Ent *t = calloc(65536, sizeof(Ent));
    uint32_t code = 0;
    for (int len = 1; len <= 16; len++) {
        for (int k = offsets[len - 1]; k < offsets[len]; k++) {
            if (code >> len) break;                                  // over-subscribed: ignored here
            uint32_t lo = code << (16 - len), hi = lo + (1u << (16 - len));
            for (uint32_t i = lo; i < hi; i++) if (!t[i].len) { t[i].len = (uint8_t)len; t[i].sym = symbols[k]; }
            code++;
        }
        code <<= 1;
    }
    return t;
}

static inline uint32_t peek16(const Sim *s, long p)
{
    long b = p >> 3;
    uint32_t v = 0;
    long nbytes = (s->nbits + 7) >> 3;
    for (int i = 0; i < 3; i++) v = (v << 8) | (b + i < nbytes ? s->ecs[b + i] : 0);
    return (v >> (8 - (p & 7))) & 0xffffu;
}

static inline int comp_of(const Sim *s, int c) { return c < s->nluma ? 0 : c - s->nluma + 1; }

// one symbol.  returns 0, or 1 if the stream's end was passed.  *ndu counts completed units.
static inline int step(const Sim *s, long *p, int *z, int *c, long *ndu, int speculative)
{
    const int k = comp_of(s, *c);
    const Ent e = (*z == 0 ? s->dc[k] : s->ac[k])[peek16(s, *p)];
    int done = 0, odd = 0;
    if (!e.len) { *p += 16; *z += 1; odd = 1; if (*z > 63) done = 1; }          // as the GPU's LUT_BAD: 16 bits, one slot
    else if (*z == 0) { int size = e.sym > 11 ? 0 : e.sym; *p += e.len + size; *z = 1; }
    else {
        const int run = e.sym >> 4, size = (e.sym & 15) > 10 ? 0 : (e.sym & 15);
        *p += e.len + size;
        if (e.sym == 0) done = 1;
        else {
            if (*z + run > 63) odd = 1;                                            // run past slot 63
            *z += run + 1;
            if (*z > 63) done = 1;
        }
    }
    if (done) {
        *z = 0; (*ndu)++;
        int adv = 1;
        if (odd && speculative) {
            if (s->policy >= 1 && s->policy <= 5) adv = 1 + s->policy;            // shift the phase
            else if (s->policy == 10) {                                             // class-aware: jump to the first unit of the OTHER class
                const int is_luma = *c < s->nluma;
                adv = is_luma ? (s->nluma - *c) : (s->dus - *c);
            }
        }
        *c = (*c + adv) % s->dus;
    }
    return *p >= s->nbits;
}

Sim *sim_open(const uint8_t *ecs, long nbytes, const uint8_t *dc_off, const uint8_t *dc_sym, const uint8_t *ac_off, const uint8_t *ac_sym,
              const int *dc_id, const int *ac_id, int nluma, int dus)
{
    Sim *s = calloc(1, sizeof(Sim));
    s->ecs = ecs; s->nbits = nbytes * 8; s->nluma = nluma; s->dus = dus;
    for (int k = 0; k < 3; k++) {
        s->dc[k] = build(dc_off + 17 * dc_id[k], dc_sym + 162 * dc_id[k]);
        s->ac[k] = build(ac_off + 17 * ac_id[k], ac_sym + 162 * ac_id[k]);
    }
    s->tz = calloc(s->nbits + 64, 1); s->tc = calloc(s->nbits + 64, 1);
    long p = 0, ndu = 0; int z = 0, c = 0;
    while (p < s->nbits) { s->tz[p] = (uint8_t)(z + 1); s->tc[p] = (uint8_t)c; if (step(s, &p, &z, &c, &ndu, 0)) break; }
    return s;
}
void sim_close(Sim *s) { for (int k = 0; k < 3; k++) { free(s->dc[k]); free(s->ac[k]); } free(s->tz); free(s->tc); free(s); }
void sim_policy(Sim *s, int policy) { s->policy = policy; }

// bits from p0 until a decoder started at (p0, DC expected, phase h) is in step with the true decoder; -1: not within max_bits
long sim_sync_distance(const Sim *s, long p0, int h, long max_bits)
{
    long p = p0, ndu = 0; int z = 0, c = h;
    while (p < s->nbits && p - p0 < max_bits) {
        if (s->tz[p] == z + 1 && s->tc[p] == c) return p - p0;
        if (step(s, &p, &z, &c, &ndu, 1)) break;
    }
    return -1;
}

// The rounds of one picture cut into subsequences of S bytes, 64 per wave (no walker, waves stitched by generations as on the GPU
// is not modelled: lane 0 of a wave takes the predecessor's exit of the same round).  out[0] = rounds of the slowest wave,
// out[1] = sum over waves of rounds, out[2] = waves, out[3] = total active lane-passes, out[4..4+32) = active lanes per round (summed over waves)
typedef struct { long p; int z, c; } St;
static St run_lane(const Sim *s, St in, long end_bit)
{
    long ndu = 0;
    while (in.p < end_bit && in.p < s->nbits) if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
    return in;
}
void sim_rounds(const Sim *s, int S, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    St *ex = calloc(nl, sizeof(St));
    uint8_t *chg = calloc(nl + 1, 1), *nchg = calloc(nl + 1, 1);
    for (int k = 0; k < nl; k++) { St in = { (long)k * S * 8, 0, 0 }; ex[k] = run_lane(s, in, (long)(k + 1) * S * 8); chg[k] = 1; }
    memset(out, 0, 40 * sizeof(long));
    const int nw = (nl + 63) / 64;
    long *wr = calloc(nw, sizeof(long));
    for (int round = 1; round < 4096; round++) {
        int any = 0;
        memset(nchg, 0, nl + 1);
        St *nex = malloc(nl * sizeof(St));
        memcpy(nex, ex, nl * sizeof(St));
        for (int k = 1; k < nl; k++) {
            if (!chg[k - 1]) continue;
            St e = run_lane(s, ex[k - 1], (long)(k + 1) * S * 8);
            out[3]++;
            if (round - 1 < 32) out[4 + round - 1]++;
            wr[k / 64] = round;
            if (e.p != ex[k].p || e.z != ex[k].z || e.c != ex[k].c) { nex[k] = e; nchg[k] = 1; any = 1; }
        }
        memcpy(ex, nex, nl * sizeof(St)); free(nex);
        memcpy(chg, nchg, nl + 1);
        if (!any) break;
    }
    for (int w = 0; w < nw; w++) { if (wr[w] > out[0]) out[0] = wr[w]; out[1] += wr[w]; }
    out[2] = nw;
    free(ex); free(chg); free(nchg); free(wr);
}

// The same rounds with a MEMO per lane: every (entry state -> exit state) a lane has decoded is kept; a round in which all the
// active lanes of a wave find their new entry in their memo costs no decode.  out[0] = decode rounds of the slowest wave,
// out[1] = sum over waves of decode rounds, out[2] = waves, out[3] = lane decodes, out[4] = sum over waves of ALL rounds,
// out[5] = memo hits, out[6] = max memo entries in a lane, out[7] = all rounds of the slowest wave
#define MEMO 12
void sim_rounds_memo(const Sim *s, int S, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    St *ex = calloc(nl, sizeof(St));
    St *min_ = calloc((size_t)nl * MEMO, sizeof(St)), *mout = calloc((size_t)nl * MEMO, sizeof(St));
    int *mn = calloc(nl, sizeof(int));
    uint8_t *chg = calloc(nl + 1, 1), *nchg = calloc(nl + 1, 1);
    for (int k = 0; k < nl; k++) {
        St in = { (long)k * S * 8, 0, 0 };
        ex[k] = run_lane(s, in, (long)(k + 1) * S * 8); chg[k] = 1;
        min_[(size_t)k * MEMO] = in; mout[(size_t)k * MEMO] = ex[k]; mn[k] = 1;
    }
    memset(out, 0, 40 * sizeof(long));
    const int nw = (nl + 63) / 64;
    long *wdec = calloc(nw, sizeof(long)), *wall = calloc(nw, sizeof(long));
    for (int round = 1; round < 4096; round++) {
        int any = 0;
        memset(nchg, 0, nl + 1);
        St *nex = malloc(nl * sizeof(St));
        memcpy(nex, ex, nl * sizeof(St));
        uint8_t *wd = calloc(nw, 1), *wa = calloc(nw, 1);
        for (int k = 1; k < nl; k++) {
            if (!chg[k - 1]) continue;
            const St in = ex[k - 1];
            St e; int hit = -1;
            for (int j = 0; j < mn[k]; j++) { const St m = min_[(size_t)k * MEMO + j]; if (m.p == in.p && m.z == in.z && m.c == in.c) { hit = j; break; } }
            if (hit >= 0) { e = mout[(size_t)k * MEMO + hit]; out[5]++; }
            else {
                e = run_lane(s, in, (long)(k + 1) * S * 8);
                out[3]++; wd[k / 64] = 1;
                if (mn[k] < MEMO) { min_[(size_t)k * MEMO + mn[k]] = in; mout[(size_t)k * MEMO + mn[k]] = e; mn[k]++; }
            }
            wa[k / 64] = 1;
            if (e.p != ex[k].p || e.z != ex[k].z || e.c != ex[k].c) { nex[k] = e; nchg[k] = 1; any = 1; }
        }
        for (int w = 0; w < nw; w++) { wdec[w] += wd[w]; wall[w] += wa[w]; }
        free(wd); free(wa);
        memcpy(ex, nex, nl * sizeof(St)); free(nex);
        memcpy(chg, nchg, nl + 1);
        if (!any) break;
    }
    for (int w = 0; w < nw; w++) { if (wdec[w] > out[0]) out[0] = wdec[w]; out[1] += wdec[w]; out[4] += wall[w]; if (wall[w] > out[7]) out[7] = wall[w]; }
    for (int k = 0; k < nl; k++) if (mn[k] > out[6]) out[6] = mn[k];
    out[2] = nw;
    free(ex); free(chg); free(nchg); free(wdec); free(wall); free(min_); free(mout); free(mn);
}

// Rounds with merge detection against the checkpoints of the last `keep` trajectories of the lane (keep = 1: what the GPU does
// today; a large keep: every trajectory the lane ever decoded).  NCHK checkpoints per subsequence.  A round of a wave lasts as long
// as its longest decode: out[0] = sum over waves of (sum over rounds of the longest decode in the round, in bytes), out[1] = the same
// for the slowest wave of the picture, out[2] = waves, out[3] = bytes decoded in re-sync passes, out[4] = rounds summed over waves,
// out[5] = rounds of the slowest wave, out[6] = merges into a trajectory older than the newest
#define TRAJ 16
typedef struct { St chk[8]; St exit; int nchk; } Traj;
static St run_lane_merge(const Sim *s, St in, long start_bit, long end_bit, int nchk, Traj *T, int nt, int keep, long *len, int *merged_old, Traj *rec)
{
    long ndu = 0;
    const long span = (end_bit - start_bit) / nchk;
    int j = 1;
    long next = start_bit + span;
    rec->nchk = 0;
    const long p0 = in.p;
    while (in.p < end_bit && in.p < s->nbits) {
        if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
        if (in.p >= next && j < nchk && in.p < end_bit) {
            // state at checkpoint j
            const int lo = nt - keep < 0 ? 0 : nt - keep;
            for (int t = nt - 1; t >= lo; t--)
                if (T[t].nchk >= j && T[t].chk[j - 1].p == in.p && T[t].chk[j - 1].z == in.z && T[t].chk[j - 1].c == in.c) {
                    *len = in.p - p0;
                    if (t != nt - 1) (*merged_old)++;
                    // the rest of the trajectory is the old one's
                    for (int q = j - 1; q < T[t].nchk; q++) rec->chk[q] = T[t].chk[q];
                    rec->nchk = T[t].nchk; rec->exit = T[t].exit;
                    return T[t].exit;
                }
            rec->chk[j - 1] = in; rec->nchk = j;
            j++; next += span;
        }
    }
    *len = in.p - p0;
    rec->exit = in;
    return in;
}
void sim_rounds_keep(const Sim *s, int S, int nchk, int keep, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    St *ex = calloc(nl, sizeof(St));
    Traj *T = calloc((size_t)nl * TRAJ, sizeof(Traj));
    int *nt = calloc(nl, sizeof(int));
    uint8_t *chg = calloc(nl + 1, 1), *nchg = calloc(nl + 1, 1);
    memset(out, 0, 40 * sizeof(long));
    int dummy = 0; long len;
    for (int k = 0; k < nl; k++) {
        St in = { (long)k * S * 8, 0, 0 };
        ex[k] = run_lane_merge(s, in, (long)k * S * 8, (long)(k + 1) * S * 8, nchk, T + (size_t)k * TRAJ, 0, keep, &len, &dummy, T + (size_t)k * TRAJ);
        nt[k] = 1; chg[k] = 1;
    }
    const int nw = (nl + 63) / 64;
    long *wt = calloc(nw, sizeof(long)), *wr = calloc(nw, sizeof(long));
    for (int round = 1; round < 4096; round++) {
        int any = 0;
        memset(nchg, 0, nl + 1);
        St *nex = malloc(nl * sizeof(St));
        memcpy(nex, ex, nl * sizeof(St));
        long *wmax = calloc(nw, sizeof(long));
        for (int k = 1; k < nl; k++) {
            if (!chg[k - 1]) continue;
            Traj rec; int mo = 0;
            Traj *Tk = T + (size_t)k * TRAJ;
            St e = run_lane_merge(s, ex[k - 1], (long)k * S * 8, (long)(k + 1) * S * 8, nchk, Tk, nt[k], keep, &len, &mo, &rec);
            out[6] += mo;
            out[3] += len / 8;
            if (len / 8 > wmax[k / 64]) wmax[k / 64] = len / 8;
            if (nt[k] < TRAJ) Tk[nt[k]++] = rec; else { memmove(Tk, Tk + 1, (TRAJ - 1) * sizeof(Traj)); Tk[TRAJ - 1] = rec; }
            if (e.p != ex[k].p || e.z != ex[k].z || e.c != ex[k].c) { nex[k] = e; nchg[k] = 1; any = 1; }
        }
        for (int w = 0; w < nw; w++) if (wmax[w]) { wt[w] += wmax[w]; wr[w]++; }
        free(wmax);
        memcpy(ex, nex, nl * sizeof(St)); free(nex);
        memcpy(chg, nchg, nl + 1);
        if (!any) break;
    }
    for (int w = 0; w < nw; w++) { out[0] += wt[w]; if (wt[w] > out[1]) out[1] = wt[w]; out[4] += wr[w]; if (wr[w] > out[5]) out[5] = wr[w]; }
    out[2] = nw;
    free(ex); free(T); free(nt); free(chg); free(nchg); free(wt); free(wr);
}

// The MULTI-HYPOTHESIS scheme: every subsequence is decoded from its first bit under every start phase h (DC expected, phase h),
// leaving nchk checkpoints each; then every exit (k, h) is continued into subsequence k + 1 until it equals a checkpoint of one of
// (k + 1, h'): a LINK (k, h) -> h' of some length.  The true path follows the links from (0, 0).
// out[0] = links, out[1] = links that found no trajectory inside the next subsequence, out[2] = sum of link lengths (bytes),
// out[3] = TRUE-path links, out[4] = true-path links without a merge, out[5] = sum of true-path link lengths,
// out[6] = lanes, out[7] = max link length on the true path, out[8..8+nchk] = histogram: checkpoint index at which links merged (all links)
// out[20] = number of distinct exit states among the hypotheses of a lane, summed over lanes
void sim_hyps(const Sim *s, int S, int nchk, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S), H = s->dus;
    Traj *T = calloc((size_t)nl * H, sizeof(Traj));
    memset(out, 0, 40 * sizeof(long));
    int dummy = 0; long len;
    for (int k = 0; k < nl; k++)
        for (int h = 0; h < H; h++) {
            St in = { (long)k * S * 8, 0, h };
            Traj *t = T + (size_t)k * H + h;
            run_lane_merge(s, in, (long)k * S * 8, (long)(k + 1) * S * 8, nchk, t, 0, 1, &len, &dummy, t);
        }
    for (int k = 0; k < nl; k++) {
        int distinct = 0;
        for (int h = 0; h < H; h++) {
            int dup = 0;
            for (int g = 0; g < h; g++) { const St a = T[(size_t)k * H + h].exit, b = T[(size_t)k * H + g].exit; if (a.p == b.p && a.z == b.z && a.c == b.c) dup = 1; }
            distinct += !dup;
        }
        out[20] += distinct;
    }
    // links
    int *link = malloc((size_t)nl * H * sizeof(int));
    long *llen = malloc((size_t)nl * H * sizeof(long));
    for (int k = 0; k + 1 < nl; k++)
        for (int h = 0; h < H; h++) {
            St in = T[(size_t)k * H + h].exit;
            const long start_bit = (long)(k + 1) * S * 8, end_bit = (long)(k + 2) * S * 8, span = (end_bit - start_bit) / nchk;
            long ndu = 0, next = start_bit + span; int j = 1, found = -1;
            const long p0 = in.p;
            while (in.p < end_bit && in.p < s->nbits && found < 0) {
                if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
                if (in.p >= next && j < nchk && in.p < end_bit) {
                    for (int g = 0; g < H; g++) { const Traj *t = T + (size_t)(k + 1) * H + g; if (t->nchk >= j && t->chk[j - 1].p == in.p && t->chk[j - 1].z == in.z && t->chk[j - 1].c == in.c) { found = g; break; } }
                    if (found >= 0) { out[8 + j]++; break; }
                    j++; next += span;
                }
            }
            if (found < 0) {      // the exit itself may equal a hypothesis' exit
                for (int g = 0; g < H; g++) { const St e = T[(size_t)(k + 1) * H + g].exit; if (e.p == in.p && e.z == in.z && e.c == in.c) { found = g; break; } }
                if (found >= 0) out[8 + nchk]++;
            }
            link[(size_t)k * H + h] = found; llen[(size_t)k * H + h] = (in.p - p0) / 8;
            out[0]++; out[2] += (in.p - p0) / 8;
            if (found < 0) out[1]++;
        }
    // the true path
    int h = 0;
    for (int k = 0; k + 1 < nl; k++) {
        const int g = link[(size_t)k * H + h];
        out[3]++; out[5] += llen[(size_t)k * H + h];
        if (llen[(size_t)k * H + h] > out[7]) out[7] = llen[(size_t)k * H + h];
        if (g < 0) { out[4]++; // walk on from the true state: find where truth is at the end of k+1 and which hypothesis of k+2 it joins: here simply restart from the true trajectory
            // truth at the start of lane k+2: follow the true decoder
            h = -1;
            // find a hypothesis of lane k+1 whose exit equals the true state at the end of lane k+1 (none by construction) -> continue with a fresh search
            St in = T[(size_t)k * H + (link[(size_t)k * H + 0] >= 0 ? 0 : 0)].exit; (void)in;
            // give up tracking exactly: resynchronise the bookkeeping by brute force
            long pe = (long)(k + 2) * S * 8;
            // true state at first symbol start >= pe
            long p = pe; while (p < s->nbits && !s->tz[p]) p++;
            for (int g2 = 0; g2 < H && k + 2 < nl; g2++) { /* which hypothesis of k+2 does truth join? unknown until linked; approximate by the hypothesis whose exit is on the true path */
                const St e = T[(size_t)(k + 2) * H + g2].exit; if (e.p < s->nbits && s->tz[e.p] == e.z + 1 && s->tc[e.p] == e.c) { h = g2; break; } }
            if (h < 0) h = 0;
            k++;    // lane k+1 was walked through
            continue;
        }
        h = g;
    }
    out[6] = nl;
    free(T); free(link); free(llen);
}

// For hypothesis sets given as bit masks: in how many lanes is the TRUE state at the lane's end equal to the exit of one of the
// set's hypotheses (started at the lane's first bit, DC expected)?  out[i] = count for masks[i]; returns the number of lanes (without lane 0).
long sim_hyp_sets(const Sim *s, int S, const int *masks, int nmasks, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S), H = s->dus;
    for (int i = 0; i < nmasks; i++) out[i] = 0;
    long n = 0;
    for (int k = 1; k < nl; k++) {
        const long end_bit = (long)(k + 1) * S * 8;
        if (end_bit >= s->nbits) break;
        long pt = end_bit; while (pt < s->nbits && !s->tz[pt]) pt++;         // first true symbol start at or behind the lane's end
        int ok[8] = {0};
        for (int h = 0; h < H; h++) {
            St in = { (long)k * S * 8, 0, h };
            const St e = run_lane(s, in, end_bit);
            ok[h] = e.p == pt && s->tz[pt] == e.z + 1 && s->tc[pt] == e.c;
        }
        for (int i = 0; i < nmasks; i++) { int any = 0; for (int h = 0; h < H; h++) if ((masks[i] >> h) & 1) any |= ok[h]; out[i] += any; }
        n++;
    }
    return n;
}

// The GPU's rounds with ONE extra cached state per lane ("B"): the state an older trajectory had at the FIRST checkpoint, with the units
// that follow it and that trajectory's exit state.  A pass that crosses checkpoint 1 in state B ends there (exit = B's exit) and the
// roles swap; a pass that crosses it merging with neither moves the newest trajectory's first checkpoint (and exit) into B.
// spec2 >= 0: the speculative pass is followed by a second one from start phase `spec2`, which fills B.
// out[] as sim_rounds_keep (+ out[7] = bytes decoded in the second speculative pass)
typedef struct { St st; St exit; int valid; } BCache;
void sim_rounds_b(const Sim *s, int S, int nchk, int spec2, int use_b, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    St *ex = calloc(nl, sizeof(St));
    Traj *A = calloc(nl, sizeof(Traj));
    BCache *Bc = calloc(nl, sizeof(BCache));
    uint8_t *chg = calloc(nl + 1, 1), *nchg = calloc(nl + 1, 1);
    memset(out, 0, 40 * sizeof(long));
    int dummy = 0; long len;
    for (int k = 0; k < nl; k++) {
        St in = { (long)k * S * 8, 0, 0 };
        ex[k] = run_lane_merge(s, in, (long)k * S * 8, (long)(k + 1) * S * 8, nchk, A + k, 0, 1, &len, &dummy, A + k);
        chg[k] = 1;
        if (spec2 >= 0 && use_b) {
            Traj t2; St in2 = { (long)k * S * 8, 0, spec2 % s->dus };
            // stops early if it meets the first trajectory at a checkpoint
            St e2 = run_lane_merge(s, in2, (long)k * S * 8, (long)(k + 1) * S * 8, nchk, A + k, 1, 1, &len, &dummy, &t2);
            out[7] += len / 8;
            if (t2.nchk >= 1 && !(t2.chk[0].p == A[k].chk[0].p && t2.chk[0].z == A[k].chk[0].z && t2.chk[0].c == A[k].chk[0].c)) { Bc[k].st = t2.chk[0]; Bc[k].exit = e2; Bc[k].valid = 1; }
        }
    }
    const int nw = (nl + 63) / 64;
    long *wt = calloc(nw, sizeof(long)), *wr = calloc(nw, sizeof(long));
    for (int round = 1; round < 4096; round++) {
        int any = 0;
        memset(nchg, 0, nl + 1);
        St *nex = malloc(nl * sizeof(St));
        memcpy(nex, ex, nl * sizeof(St));
        long *wmax = calloc(nw, sizeof(long));
        for (int k = 1; k < nl; k++) {
            if (!chg[k - 1]) continue;
            // decode from the predecessor's exit; at checkpoint 1 also compare with B
            St in = ex[k - 1];
            const long start_bit = (long)k * S * 8, end_bit = (long)(k + 1) * S * 8, span = (end_bit - start_bit) / nchk;
            long ndu = 0, next = start_bit + span; int j = 1;
            const long p0 = in.p;
            Traj rec; rec.nchk = 0;
            St e; int done = 0;
            const St oldA1 = A[k].chk[0]; const int oldA_has1 = A[k].nchk >= 1; const St old_exit = ex[k];
            while (in.p < end_bit && in.p < s->nbits && !done) {
                if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
                if (in.p >= next && j < nchk && in.p < end_bit) {
                    if (A[k].nchk >= j && A[k].chk[j - 1].p == in.p && A[k].chk[j - 1].z == in.z && A[k].chk[j - 1].c == in.c) {
                        for (int q = j - 1; q < A[k].nchk; q++) rec.chk[q] = A[k].chk[q];
                        rec.nchk = A[k].nchk; e = A[k].exit; done = 1; break;
                    }
                    if (use_b && j == 1 && Bc[k].valid && Bc[k].st.p == in.p && Bc[k].st.z == in.z && Bc[k].st.c == in.c) {
                        rec.chk[0] = in; rec.nchk = 1; e = Bc[k].exit; done = 2; out[6]++; break;
                    }
                    rec.chk[j - 1] = in; rec.nchk = j;
                    j++; next += span;
                }
            }
            if (!done) e = in;
            len = in.p - p0;
            rec.exit = e;
            // B takes over the newest trajectory's first checkpoint when the pass crossed checkpoint 1 without merging into A there
            if (use_b && oldA_has1 && rec.nchk >= 1 && !(rec.chk[0].p == oldA1.p && rec.chk[0].z == oldA1.z && rec.chk[0].c == oldA1.c)) { Bc[k].st = oldA1; Bc[k].exit = old_exit; Bc[k].valid = 1; }
            A[k] = rec;
            out[3] += len / 8;
            if (len / 8 > wmax[k / 64]) wmax[k / 64] = len / 8;
            if (e.p != ex[k].p || e.z != ex[k].z || e.c != ex[k].c) { nex[k] = e; nchg[k] = 1; any = 1; }
        }
        for (int w = 0; w < nw; w++) if (wmax[w]) { wt[w] += wmax[w]; wr[w]++; }
        free(wmax);
        memcpy(ex, nex, nl * sizeof(St)); free(nex);
        memcpy(chg, nchg, nl + 1);
        if (!any) break;
    }
    for (int w = 0; w < nw; w++) { out[0] += wt[w]; if (wt[w] > out[1]) out[1] = wt[w]; out[4] += wr[w]; if (wr[w] > out[5]) out[5] = wr[w]; }
    out[2] = nw;
    free(ex); free(A); free(Bc); free(chg); free(nchg); free(wt); free(wr);
}

// QUANTUM rounds: a round lasts one checkpoint interval.  Every lane is a small machine: IDLE, or RUNNING a decode from some entry
// state towards its subsequence end, one checkpoint interval per quantum.  At a checkpoint crossing the state is compared with the
// newest trajectory's (A) and, at checkpoint 1, with the cached B: a match ends the run (the exit is known).  A run that ends with a
// NEW exit restarts its successor from it in the next quantum (a successor still running from an older entry is restarted).
// out[0] = sum over waves of quanta, out[1] = quanta of the slowest wave, out[2] = waves, out[3] = lane-quanta executed (work),
// out[4] = bytes decoded
typedef struct { int running; St cur; long ndu; int j; Traj rec; St entry; } Mach;
void sim_quanta(const Sim *s, int S, int nchk, int use_b, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    St *ex = calloc(nl, sizeof(St));
    Traj *A = calloc(nl, sizeof(Traj));
    BCache *Bc = calloc(nl, sizeof(BCache));
    Mach *M = calloc(nl, sizeof(Mach));
    memset(out, 0, 40 * sizeof(long));
    int dummy = 0; long len;
    for (int k = 0; k < nl; k++) {
        St in = { (long)k * S * 8, 0, 0 };
        ex[k] = run_lane_merge(s, in, (long)k * S * 8, (long)(k + 1) * S * 8, nchk, A + k, 0, 1, &len, &dummy, A + k);
    }
    // after the speculative pass every lane but the first of the picture is started from its predecessor's exit
    for (int k = 1; k < nl; k++) { M[k].running = 1; M[k].cur = ex[k - 1]; M[k].entry = ex[k - 1]; M[k].ndu = 0; M[k].j = 1; M[k].rec.nchk = 0; }
    const int nw = (nl + 63) / 64;
    long *wq = calloc(nw, sizeof(long));
    for (int q = 1; q < 100000; q++) {
        int any = 0;
        uint8_t *wa = calloc(nw, 1);
        St *newexit = malloc(nl * sizeof(St)); uint8_t *has = calloc(nl, 1);
        for (int k = 1; k < nl; k++) {
            if (!M[k].running) continue;
            any = 1; wa[k / 64] = 1; out[3]++;
            Mach *m = &M[k];
            const long start_bit = (long)k * S * 8, end_bit = (long)(k + 1) * S * 8, span = (end_bit - start_bit) / nchk;
            const long next = start_bit + span * m->j;
            const long p0 = m->cur.p;
            int fin = 0; St e;
            // decode until the next checkpoint boundary (or the end)
            while (m->cur.p < end_bit && m->cur.p < s->nbits) {
                if (step(s, &m->cur.p, &m->cur.z, &m->cur.c, &m->ndu, 1)) break;
                if (m->cur.p >= next && m->j < nchk) break;
            }
            out[4] += (m->cur.p - p0) / 8;
            if (m->cur.p >= end_bit || m->cur.p >= s->nbits || m->j >= nchk) {
                if (m->cur.p >= end_bit || m->cur.p >= s->nbits) { fin = 1; e = m->cur; }
            }
            if (!fin) {
                const int j = m->j;
                if (A[k].nchk >= j && A[k].chk[j - 1].p == m->cur.p && A[k].chk[j - 1].z == m->cur.z && A[k].chk[j - 1].c == m->cur.c) {
                    for (int t = j - 1; t < A[k].nchk; t++) m->rec.chk[t] = A[k].chk[t];
                    m->rec.nchk = A[k].nchk; e = A[k].exit; fin = 1;
                } else if (use_b && j == 1 && Bc[k].valid && Bc[k].st.p == m->cur.p && Bc[k].st.z == m->cur.z && Bc[k].st.c == m->cur.c) {
                    m->rec.chk[0] = m->cur; m->rec.nchk = 1; e = Bc[k].exit; fin = 1;
                } else { m->rec.chk[j - 1] = m->cur; m->rec.nchk = j; m->j++; }
            }
            if (fin) {
                m->running = 0;
                m->rec.exit = e;
                if (use_b && A[k].nchk >= 1 && m->rec.nchk >= 1 && !(m->rec.chk[0].p == A[k].chk[0].p && m->rec.chk[0].z == A[k].chk[0].z && m->rec.chk[0].c == A[k].chk[0].c)) { Bc[k].st = A[k].chk[0]; Bc[k].exit = ex[k]; Bc[k].valid = 1; }
                A[k] = m->rec;
                if (e.p != ex[k].p || e.z != ex[k].z || e.c != ex[k].c) { newexit[k] = e; has[k] = 1; }
            }
        }
        for (int k = 1; k < nl; k++) if (has[k]) {
            ex[k] = newexit[k];
            if (k + 1 < nl) { Mach *m = &M[k + 1]; m->running = 1; m->cur = ex[k]; m->entry = ex[k]; m->ndu = 0; m->j = 1; m->rec.nchk = 0; }
        }
        for (int w = 0; w < nw; w++) wq[w] += wa[w];
        free(wa); free(newexit); free(has);
        if (!any) break;
    }
    for (int w = 0; w < nw; w++) { out[0] += wq[w]; if (wq[w] > out[1]) out[1] = wq[w]; }
    out[2] = nw;
    free(ex); free(A); free(Bc); free(M); free(wq);
}

// One WAVE at a time (64 lanes, the first lane's entry is the true state), as the GPU runs it: the speculative pass, rounds while more
// than walk_max lanes are active (a round lasts as long as its longest decode), then the walk (chains one after the other, WALK_COST of a
// lane's time per byte, a changed exit carried into the next lane).  Landing pads at checkpoint 1: A (newest trajectory, all checkpoints),
// B (the one before), and with nhyp > 0 up to two HYPOTHESES (start phases 2 and 4, or 1 and 2 for three-unit MCUs) decoded for the
// lanes that are active in a round with room for helpers (active <= 21, from the second round on): free, the round is a full pass anyway.
// out[0] = sum over waves of time (bytes of lane-decode equivalents), out[1] = slowest wave, out[2] = waves, out[3] = rounds, out[4] = lanes walked,
// out[5] = walked bytes, out[6] = merges into a hypothesis
#define WALK_COST 0.2
typedef struct { St st; St exit; int valid; } Pad;
static int st_eq(St a, St b) { return a.p == b.p && a.z == b.z && a.c == b.c; }
// decode lane k from `in` until it meets a landing pad or the end; returns the exit, *len = bytes decoded, updates A (and B by the swap rule)
static St lane_pass(const Sim *s, int S, int nchk, long lane0_bit, St in, Traj *A, St *cur_exit, Pad *B, Pad *H, int nh, long *len, long *hypm)
{
    const long start_bit = lane0_bit, end_bit = lane0_bit + (long)S * 8, span = (long)S * 8 / nchk;
    long ndu = 0, next = start_bit + span; int j = 1;
    const long p0 = in.p;
    Traj rec; rec.nchk = 0;
    St e; int done = 0;
    const St oldA1 = A->chk[0]; const int oldA_has1 = A->nchk >= 1; const St old_exit = *cur_exit;
    while (in.p < end_bit && in.p < s->nbits && !done) {
        if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
        if (in.p >= next && j < nchk && in.p < end_bit) {
            if (A->nchk >= j && st_eq(A->chk[j - 1], in)) { for (int q = j - 1; q < A->nchk; q++) rec.chk[q] = A->chk[q]; rec.nchk = A->nchk; e = A->exit; done = 1; break; }
            if (j == 1) {
                if (B->valid && st_eq(B->st, in)) { rec.chk[0] = in; rec.nchk = 1; e = B->exit; done = 2; break; }
                for (int h = 0; h < nh && !done; h++) if (H[h].valid && st_eq(H[h].st, in)) { rec.chk[0] = in; rec.nchk = 1; e = H[h].exit; done = 3; (*hypm)++; }
                if (done) break;
            }
            rec.chk[j - 1] = in; rec.nchk = j; j++; next += span;
        }
    }
    if (!done) e = in;
    *len = (in.p - p0) / 8;
    rec.exit = e;
    if (oldA_has1 && rec.nchk >= 1 && !st_eq(rec.chk[0], oldA1)) { B->st = oldA1; B->exit = old_exit; B->valid = 1; }
    *A = rec;
    return e;
}
void sim_waves(const Sim *s, int S, int nchk, int walk_max, int nhyp, double *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int w0 = 0; w0 < nl; w0 += 64) {
        const int n = nl - w0 < 64 ? nl - w0 : 64;
        St ex[64]; Traj A[64]; Pad B[64]; Pad H[64][2]; uint8_t chg[65], nchg[65];
        memset(B, 0, sizeof B); memset(H, 0, sizeof H);
        int dummy = 0; long len;
        double t = 0;
        for (int k = 0; k < n; k++) {
            St in = { (long)(w0 + k) * S * 8, 0, 0 };
            ex[k] = run_lane_merge(s, in, in.p, in.p + (long)S * 8, nchk, A + k, 0, 1, &len, &dummy, A + k);
            chg[k] = 1;
        }
        t += S;                                                   // the speculative pass
        // the true entry of the wave's first lane
        St truth0; { long p = (long)w0 * S * 8; while (p < s->nbits && !s->tz[p]) p++; truth0.p = p; truth0.z = s->tz[p] ? s->tz[p] - 1 : 0; truth0.c = s->tc[p]; }
        int first = 1;
        for (int round = 1; round < 200; round++) {
            int act[64], na = 0;
            for (int k = 0; k < n; k++) { const int a = k == 0 ? (first && w0 > 0) : chg[k - 1]; if (a) act[na++] = k; }
            first = 0;
            if (!na) break;
            out[3]++;
            memset(nchg, 0, sizeof nchg);
            if (na <= walk_max) {
                // the walk: chains one after the other
                uint8_t pending[64] = {0};
                for (int i = 0; i < na; i++) pending[act[i]] = 1;
                for (int k = 0; k < n; k++) {
                    if (!pending[k]) continue;
                    int a = k;
                    for (;;) {
                        pending[a] = 0;
                        const St in = a == 0 ? truth0 : ex[a - 1];
                        const St e = lane_pass(s, S, nchk, (long)(w0 + a) * S * 8, in, A + a, ex + a, B + a, H[a], nhyp, &len, (long *)&dummy);
                        out[4]++; out[5] += len; t += len * WALK_COST + 0.03 * S * WALK_COST * 5;   // + a start-up cost per walked lane (~2 us)
                        if (st_eq(e, ex[a])) break;
                        ex[a] = e;
                        if (a + 1 >= n) break;
                        a++;
                    }
                }
                break;
            }
            long longest = 0;
            St nex[64]; memcpy(nex, ex, sizeof nex);
            long hm = 0;
            for (int i = 0; i < na; i++) {
                const int k = act[i];
                const St in = k == 0 ? truth0 : ex[k - 1];
                const St e = lane_pass(s, S, nchk, (long)(w0 + k) * S * 8, in, A + k, ex + k, B + k, H[k], nhyp, &len, &hm);
                if (len > longest) longest = len;
                if (!st_eq(e, ex[k])) { nex[k] = e; nchg[k] = 1; }
            }
            out[6] += hm;
            if (nhyp && round >= 2 && na * nhyp + na <= 64) {
                // helpers: idle lanes decode the active lanes' subsequences from their first bit under other start phases
                for (int i = 0; i < na; i++) {
                    const int k = act[i];
                    for (int h = 0; h < nhyp; h++) {
                        if (H[k][h].valid) continue;
                        St in = { (long)(w0 + k) * S * 8, 0, (s->dus >= 6 ? 2 + 2 * h : 1 + h) % s->dus };
                        Traj tt; tt.nchk = 0; Traj none; none.nchk = 0;
                        St e = run_lane_merge(s, in, in.p, in.p + (long)S * 8, nchk, &none, 0, 1, &len, &dummy, &tt);
                        if (tt.nchk >= 1) { H[k][h].st = tt.chk[0]; H[k][h].exit = e; H[k][h].valid = 1; }
                    }
                }
                longest = S;
            }
            memcpy(ex, nex, sizeof nex);
            memcpy(chg, nchg, sizeof chg);
            t += longest;
        }
        out[0] += t; if (t > out[1]) out[1] = t; out[2]++;
    }
}

// ---- round 4, #31: can a speculative pass guess the MCU phase from what it decodes? ------------------------------------------
// Natural pictures: luma units are long, chroma units short.  The detector keeps the bit lengths of the last `dus` units it completed,
// each with the phase position it decoded it at, and -- every `every` units -- re-labels its phase to the cyclic shift under which the
// long units sit on luma positions (score = sum of lengths on luma positions minus `wc` x sum on chroma positions), if that shift beats
// the current labelling by `margin` bits.  sim_exit_right: lanes of S bytes started at (lane start, DC expected, phase 0): out[0] = lanes,
// out[1] = lanes whose exit state is the true one without the detector, out[2] = with it; out[3] / out[4] = sum of bits until in step
// (lanes that get there), out[5] / out[6] = lanes that get there.
typedef struct { long len[8]; int pos[8]; int n; long last_p; int since; } Det;
static void det_unit(const Sim *s, Det *d, long p_now, int pos_decoded, int *c, int every, int wc, long margin)
{
    const int dus = s->dus;
    for (int k = dus - 1; k > 0; k--) { d->len[k] = d->len[k - 1]; d->pos[k] = d->pos[k - 1]; }
    d->len[0] = p_now - d->last_p; d->pos[0] = pos_decoded; d->last_p = p_now;
    if (d->n < dus) d->n++;
    d->since++;
    if (d->n < dus || d->since < every) return;
    long best = 0, cur = 0; int bs = 0;
    for (int sh = 0; sh < dus; sh++) {
        long sc = 0;
        for (int k = 0; k < dus; k++) { const int q = (d->pos[k] + sh) % dus; sc += q < s->nluma ? d->len[k] : -(long)wc * d->len[k]; }
        if (sh == 0) cur = sc;
        if (sh == 0 || sc > best) { best = sc; bs = sh; }
    }
    if (bs != 0 && best > cur + margin) {
        *c = (*c + bs) % dus;
        for (int k = 0; k < dus; k++) d->pos[k] = (d->pos[k] + bs) % dus;
        d->since = 0;
    }
}
static St run_lane_det(const Sim *s, St in, long end_bit, int use, int every, int wc, long margin, long *sync_at)
{
    Det d; memset(&d, 0, sizeof d); d.last_p = in.p;
    long ndu = 0; *sync_at = -1;
    while (in.p < end_bit && in.p < s->nbits) {
        if (*sync_at < 0 && s->tz[in.p] == in.z + 1 && s->tc[in.p] == in.c) *sync_at = in.p;
        const long before = ndu; const int pos = in.c;
        if (step(s, &in.p, &in.z, &in.c, &ndu, 1)) break;
        if (use && ndu != before) det_unit(s, &d, in.p, pos, &in.c, every, wc, margin);
    }
    return in;
}
void sim_exit_right(const Sim *s, int S, int every, int wc, long margin, long *out)
{
    const long nbytes = s->nbits / 8;
    const int nl = (int)((nbytes + S - 1) / S);
    memset(out, 0, 8 * sizeof(long));
    for (int k = 1; k < nl; k++) {
        const long p0 = (long)k * S * 8, end = (long)(k + 1) * S * 8;
        if (end >= s->nbits) break;
        out[0]++;
        for (int use = 0; use < 2; use++) {
            St in = { p0, 0, 0 }; long sync_at;
            const St ex = run_lane_det(s, in, end, use, every, wc, margin, &sync_at);
            if (s->tz[ex.p] == ex.z + 1 && s->tc[ex.p] == ex.c) out[1 + use]++;
            if (sync_at >= 0) { out[3 + use] += sync_at - p0; out[5 + use]++; }
        }
    }
}
