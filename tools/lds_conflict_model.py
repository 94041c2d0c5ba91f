#!/usr/bin/env python3
"""Where do the balanced decoder's LDS bank conflicts come from?  A model of sqoa_decode_bal.hip's LDS accesses on REAL streams, CPU only.

The counters say `SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS` = 1.74 for decode_bal_kernel on the headline batch (254 conflict cycles and 146 active
cycles per 1 024-byte chunk); the round-5 verdict asked for the conflicts to be taken out by layout changes (8-byte op table, skewed list / pixel
region, the second table pass from class bits).  Before building any of them: which accesses make the conflicts, and are they of the kind a
layout can remove (lanes a fixed stride apart: photographs) or of the kind it cannot (64 lanes looking up 64 data-dependent table entries: a
balls-into-bins collision, whatever the stride of the table)?

The model walks the streams of a few images (class mixed = the headline batch; class photo) chunk by chunk exactly as the kernel does -- 16 bytes
per lane, the parser's states with the RUN fusion (sqoa_decode_bal.hip, the enum), op starts ranked and dealt out q = ceil(n / 64) per lane -- and
counts, per LDS instruction, the LDS-array cycles MI355X_MICROARCH.md's table gives: one cycle per lane group (2 x 32 lanes for b32 / b64, 4 x 16
for b128) plus one for every further DISTINCT address on a busy bank of the group (bank = dword mod 32 for ds_read_b32 and writes, mod 64 for
b64 / b128); stores pay only beyond 2-way.  It prints conflict cycles per chunk by access class, and next to each what the same 64 lanes
would cost if their addresses were independent uniform draws over the table (the floor no layout gets under).

usage: python tools/lds_conflict_model.py [images per class, default 2]      (needs oracle/: test infrastructure, like every tool that checks)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po

S_A, S_N, S_G, S_X, S_D1 = 0, 1, 2, 3, 4
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 += [[x + 32 for x in g] for g in G128]
G32 = [list(range(32)), list(range(32, 64))]


def cycles(addrs, active, width, write=False):
    """LDS-array cycles of one wave-instruction: addrs = 64 byte addresses, active = 64 bools, width in bytes (4, 8, 16)."""
    groups = G128 if width == 16 else G32
    nb = 32 if (width == 4 or write) else 64
    tot = base = 0
    for g in groups:
        banks = {}
        for l in g:
            if not active[l]:
                continue
            for d in range(width // 4):
                a = addrs[l] // 4 + d
                banks.setdefault(a % nb, set()).add(a)
        worst = max((len(v) for v in banks.values()), default=0)
        if write:
            worst = (worst + 1) // 2
        base += 1
        tot += max(worst, 1)
    return tot, base


def parse_chunk(b, st):
    """b: the chunk's bytes; st: parser state in front of byte 0.  -> (op start offsets, state behind the chunk)."""
    starts = []
    for k, x in enumerate(b):
        is_abs, is_luma, is_run, is_sfx = x >= 0xFE, (x & 0xC0) == 0x80, 0xC0 <= x < 0xFE, 0x60 <= x < 0x80
        ln = x - 0xFA if is_abs else 2 if is_luma else 1
        opn = S_A if ln == 1 else S_D1 + ln - 2
        if st >= S_D1:
            nxt = S_A if st == S_D1 else st - 1
        elif st == S_X:
            nxt = S_X
        else:
            nxt = {S_A: (S_N if is_sfx else S_G if is_run else opn), S_N: (S_G if is_run else opn), S_G: (S_X if is_sfx else opn)}[st]
            if nxt == S_A or nxt >= S_D1:
                starts.append(k)
        st = nxt
    return starts, st


def model(stream, acc, rng):
    body = np.frombuffer(stream, dtype=np.uint8)[15:-8]
    st = S_G
    for c0 in range(0, len(body) - 1024, 1024):   # interior chunks
        b = body[c0 : c0 + 1024]
        starts, st = parse_chunk(b.tolist(), st)
        if st == S_X:
            return
        n = len(starts)
        if n == 0 or n > 512:
            continue
        acc["chunks"] += 1
        acc["ops"] += n
        mis = (15 + c0) & 3
        on = [True] * 64
        # the transition table: 16 look-ups per lane, twice (parse, starts pass): s_ftab[byte], 4-byte entries
        for k in range(16):
            t, bs = cycles([int(b[16 * l + k]) * 4 for l in range(64)], on, 4)
            acc["ftab (2 passes x 16 reads)"] += 2 * np.array([t - bs, bs])
            t, bs = cycles([int(x) * 4 for x in rng.integers(0, 256, 64)], on, 4)
            acc["ftab (2 passes x 16 reads)|floor"] += 2 * (t - bs)
        # parse: the lane's 16 staged bytes + the dword behind them
        t, bs = cycles([16 * l for l in range(64)], on, 16)
        acc["staged bytes (parse read, stage write)"] += np.array([t - bs, bs])
        t, bs = cycles([16 * l + 16 for l in range(64)], on, 4)
        acc["staged bytes (parse read, stage write)"] += np.array([t - bs, bs])
        # the list: written by the byte owners at rank, read by the op owners
        per_lane = [[s for s in starts if s // 16 == l] for l in range(64)]
        rank0 = np.concatenate([[0], np.cumsum([len(p) for p in per_lane])[:-1]])
        for i in range(8):
            act = [len(p) > i for p in per_lane]
            if any(act):
                t, bs = cycles([4 * (8 + int(rank0[l]) + i) for l in range(64)], act, 4, write=True)
                acc["start list (writes by rank, reads q apart)"] += np.array([t - bs, bs])
        q = (n + 63) // 64
        k = n - 64 * (q - 1)
        r0 = [l * (q - 1) + min(l, k) - (0 if l < k else 1) for l in range(64)]
        for i in range(0, 8, 2):   # ds_read2_b32 x 4: two dword accesses each
            for j in (i, i + 1):
                t, bs = cycles([4 * (8 + r0[l] + j) for l in range(64)], on, 4)
                acc["start list (writes by rank, reads q apart)"] += np.array([t - bs, bs])
        # pixels per walk iteration (the op, its ALPHA suffix, a RUN taken along: sqoa_decode_bal.hip, walk) -> every op's pixel position in the chunk
        def runcnt(x):
            return 512 if x == 0xFD else (x & 0x3F) + 1

        cnt = []
        for sp in starts:
            x = int(b[sp])
            is_run = 0xC0 <= x < 0xFE
            p2 = sp + (x - 0xFA if x >= 0xFE else 2 if (x & 0xC0) == 0x80 else 1)
            c1 = runcnt(x) if is_run else 1
            if p2 < 1024 and 0x60 <= int(b[p2]) < 0x80:
                p2 += 1
            if p2 < 1024 and 0xC0 <= int(b[p2]) < 0xFE:
                c1 += runcnt(int(b[p2]))
            cnt.append(c1)
        pxpos = np.concatenate([[0], np.cumsum(cnt)])
        # the walk: per slot three window dwords, the op table entry (16 B), two suffix-table dwords
        for i in range(q):
            act = [(i > 0 or l < k) for l in range(64)]
            pos = [starts[min(n - 1, max(0, r0[l] + i))] for l in range(64)]
            for d in range(3):
                t, bs = cycles([((pos[l] + mis) & ~3) + 4 * d for l in range(64)], act, 4)
                acc["windows (3 dwords per op)"] += np.array([t - bs, bs])
            opb = [int(b[p]) for p in pos]
            t, bs = cycles([16 * x for x in opb], act, 16)
            acc["op table (ds_read_b128)"] += np.array([t - bs, bs])
            t, bs = cycles([16 * int(x) for x in rng.choice(np.array(opb)[np.array(act)], 64)] if any(act) else [0] * 64, act, 16)
            acc["op table (ds_read_b128)|floor"] += t - bs
            for _ in range(2):   # the byte behind the op, and behind its suffix: mostly the next op's first byte
                nb = [int(b[min(1023, p + (4 if x == 0xFE else 5 if x == 0xFF else 2 if (x & 0xC0) == 0x80 else 1))]) for p, x in zip(pos, opb)]
                t, bs = cycles([4 * x for x in nb], act, 4)
                acc["suffix / RUN table (2 reads per op)"] += np.array([t - bs, bs])
            # the pixel region: one dword per op at its pixel position in the chunk (+ the second pixel of pairs: same banks + 1)
            t, bs = cycles([4 * int(pxpos[min(n - 1, max(0, r0[l] + i))] % 768) for l in range(64)], act, 4, write=True)
            acc["pixel region (one write per op)"] += np.array([t - bs, bs])


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rng = np.random.default_rng(6)
    o = po.best()
    for cls in ("mixed", "photo"):
        acc = {}

        class Acc(dict):
            def __missing__(self, key):
                self[key] = 0 if key in ("chunks", "ops") or key.endswith("|floor") else np.zeros(2)
                return self[key]

        acc = Acc()
        for img in range(n_img):
            px = po.synth(1024, 1024, 4, cls, 1, img)
            stream = o.encode(px, 1024, 1024, 4, 0)
            model(stream[: 15 + 1024 * 400 + 8], acc, rng)   # the first 400 chunks of each image
        ch = acc["chunks"]
        print(f"class {cls}: {ch} chunks of {n_img} image(s), {acc['ops'] / ch:.0f} op starts per chunk (q = {int(np.ceil(acc['ops'] / ch / 64))})")
        print(f"  {'access class':<52} {'conflict cycles':>16} {'conflict-free':>14}   {'random-draw floor':>18}")
        tc = tb = 0.0
        for key, v in acc.items():
            if key in ("chunks", "ops") or key.endswith("|floor"):
                continue
            fl = acc.get(key + "|floor")
            print(f"  {key:<52} {v[0] / ch:>16.1f} {v[1] / ch:>14.1f}   {'' if fl is None else format(fl / ch, '18.1f')}")
            tc += v[0] / ch
            tb += v[1] / ch
        print(f"  {'sum (per chunk and wave)':<52} {tc:>16.1f} {tb:>14.1f}   ratio {tc / tb:.2f}   (counters, headline batch: 254 / 146 = 1.74)")


if __name__ == "__main__":
    main()
