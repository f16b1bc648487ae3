"""Eager two-stream step vs replayed hipGraph step from two rocprofv3 kernel traces (rocpd sqlite):
    python scripts/prof_timeline2.py <eager.db> <graph.db> [steps_from_end]
For the last full step of each trace (steps are cut at the last clip_adam_kernel launch of a group): wall time from the first kernel start to the last
kernel end, the union of the kernel intervals (= time at least one kernel runs), the idle remainder, the overlap (sum - union), launches, and the
phases (cut at marker kernels) with their wall / union / idle -- the place where the two launch paths differ is where a phase's idle time differs."""
import re
import sqlite3
import sys


def load(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    tcol = "start" if "start" in cols else "start_timestamp"
    ecol = "end" if "end" in cols else "end_timestamp"
    extra = [c for c in ("stream_id", "queue_id") if c in cols]
    rows = con.execute(f"select name, {tcol}, {ecol}{''.join(', ' + c for c in extra)} from kernels order by {tcol}").fetchall()
    return rows, extra


def last_step(rows, back=1):
    ends = [i for i, r in enumerate(rows) if "clip_adam" in r[0] and not any("clip_adam" in rows[j][0] for j in range(i + 1, min(i + 4, len(rows))))]
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    return rows[lo:hi]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


PHASES = [("backbone fwd", None), ("encoder fwd", "attn2_fwd|attn_fwd"), ("set loss", "match_cost"), ("backward", "set_loss_grad"), ("optimiser", "sumsq_segments")]


def report(tag, step, extra):
    t0, t1 = step[0][1], max(r[2] for r in step)
    iv = [(r[1], r[2]) for r in step]
    u = union(iv)
    s = sum(e - b for b, e in iv)
    print(f"== {tag}: {len(step)} launches, wall {(t1 - t0) / 1e3:.1f} us, kernel sum {s / 1e3:.1f} us, union {u / 1e3:.1f} us, idle {(t1 - t0 - u) / 1e3:.1f} us, overlap {(s - u) / 1e3:.1f} us")
    if extra:
        ids = {}
        for r in step:
            ids.setdefault(r[3], []).append((r[1], r[2]))
        for k, v in sorted(ids.items(), key=lambda kv: -len(kv[1])):
            print(f"   {extra[0]} {k}: {len(v)} launches, busy {union(v) / 1e3:.1f} us")
    # phases: cut at the first launch matching each marker
    cuts = [0]
    for name, pat in PHASES[1:]:
        idx = next((i for i, r in enumerate(step) if i > cuts[-1] and re.search(pat, r[0])), None)
        cuts.append(idx if idx is not None else cuts[-1])
    cuts.append(len(step))
    for (name, _), a, b in zip(PHASES, cuts[:-1], cuts[1:]):
        seg = step[a:b]
        if not seg:
            continue
        w0, w1 = seg[0][1], max(r[2] for r in seg)
        uu = union([(r[1], r[2]) for r in seg])
        print(f"   {name:14s} {len(seg):4d} launches  wall {(w1 - w0) / 1e3:8.1f}  union {uu / 1e3:8.1f}  idle {(w1 - w0 - uu) / 1e3:7.1f} us")
    # the largest idle gaps
    ivs = sorted(iv)
    gaps, ce = [], ivs[0][1]
    for i, (b, e) in enumerate(ivs[1:], 1):
        if b > ce:
            gaps.append((b - ce, i))
        ce = max(ce, e)
    srt = sorted(step, key=lambda r: r[1])
    short = lambda n: re.sub(r"\s+", " ", re.sub(r"\(.*", "", n)).replace("void detr::", "").replace("detr::", "")[:60]
    print("   largest idle gaps (us, before launch #, kernel): " + "; ".join(f"{g / 1e3:.1f} #{i} {short(srt[i][0])}" for g, i in sorted(gaps, reverse=True)[:8]))
    print(f"   gaps > 2 us: {sum(1 for g, _ in gaps if g > 2000)} totalling {sum(g for g, _ in gaps if g > 2000) / 1e3:.1f} us; gaps <= 2 us: {sum(1 for g, _ in gaps if g <= 2000)} totalling {sum(g for g, _ in gaps if g <= 2000) / 1e3:.1f} us")


back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for tag, db in (("eager, two streams", sys.argv[1]), ("hipGraph replay", sys.argv[2])):
    rows, extra = load(db)
    report(tag, last_step(rows, back), extra)
