"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite) per kernel name.
usage: python scripts/pmc_summary.py <out_dir> [<name substring> <traffic.json>]"""
import glob
import json
import re
import sqlite3
import sys


def fold(name):
    """rocprofv3 kernel name -> the family name detr_tf/_hip.py reports: one kernel BODY with every template instantiation
    pooled (layouts, storage types, tile sizes, K-tile depths; grouped launches folded into their base kernel).  ONE rule for
    bench.py, this script and DESIGN (VERDICT r4 #9a): the bf16 tile-GEMM family is gemm_bf16c{,_group,_k64,_ln}_kernel AND the
    round-5 ring kernel gemm_ring_kernel -- every launch that detr_tf/_hip.py bills to "gemm_bf16c_kernel"."""
    m = re.search(r"detr::(gemm_(?:bf16c|f32))(?:_group|_k64|_ln)?_kernel<", name)
    if m:
        return f"{m.group(1)}_kernel"
    if "detr::gemm_ring_kernel" in name:
        return "gemm_bf16c_kernel"
    if "detr::gemm_stream_bf16_kernel" in name:
        return "gemm_stream_bf16_kernel"
    return None


def main(argv):
    out = argv[1]
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs = glob.glob(f"{out}/pmc_{c}/**/*.db", recursive=True)
        if not dbs:
            print("no db for", c)
            continue
        con = sqlite3.connect(dbs[0])
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        print("#", c, "tables/views:", [t for t in tabs if "counter" in t.lower() or "pmc" in t.lower()][:12])
        view = "counters_collection" if "counters_collection" in tabs else None
        if view is None:
            continue
        cols = [d[1] for d in con.execute(f"pragma table_info({view})")]
        print("#", view, cols)
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {namecol}, counter_name, count(*), sum(value) from {view} group by {namecol}, counter_name"
        for name, cn, n, v in con.execute(q):
            res.setdefault(re.sub(r"\s+", " ", name)[:110], {})[cn] = (n, v)
    rows = sorted(res.items(), key=lambda kv: -sum(x[1] for x in kv[1].values()))
    print(f"{'kernel':110s} {'launches':>8} {'FETCH_SIZE':>14} {'WRITE_SIZE':>14}   (raw counter units; see MI355X_MICROARCH.md #HBM for the gfx950 x2 read correction)")
    for k, v in rows[:40]:
        f = v.get("FETCH_SIZE", (0, 0))
        w = v.get("WRITE_SIZE", (0, 0))
        print(f"{k:110s} {max(f[0], w[0]):8d} {f[1]:14.0f} {w[1]:14.0f}")

    # optional: aggregate one kernel family into the traffic JSON bench.py reads (argv[2] = name substring, argv[3] = path)
    if len(argv) >= 4:
        pat, path = argv[2], argv[3]
        n = fetch = write = 0
        for k, v in res.items():
            if pat in k:
                f = v.get("FETCH_SIZE", (0, 0)); w = v.get("WRITE_SIZE", (0, 0))
                n += max(f[0], w[0]); fetch += f[1]; write += w[1]
        per = {}
        for k, v in res.items():
            key = fold(k)
            if key is None:
                continue
            f = v.get("FETCH_SIZE", (0, 0)); w = v.get("WRITE_SIZE", (0, 0))
            e = per.setdefault(key, {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
            e["launches"] += max(f[0], w[0]); e["FETCH_SIZE_KB"] += f[1]; e["WRITE_SIZE_KB"] += w[1]
        for e in per.values():
            e["traffic_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024.0 / max(e["launches"], 1)
        if n:
            json.dump({"kernel": pat + "*", "per_symbol": per, "launches": n, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                       "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / n,
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE separate passes over bench.py --steps 2 --warmup 1; bytes = "
                               "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md gfx950 correction; launches of all shapes pooled"},
                      open(path, "w"), indent=1)
            print("wrote", path)


if __name__ == "__main__":
    main(sys.argv)
