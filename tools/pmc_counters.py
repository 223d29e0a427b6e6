"""Developer tool: per-kernel means of arbitrary rocprofv3 --pmc counters from one or more pass .db files.

    python tools/pmc_counters.py <pass1.db> [<pass2.db> ...]   ->  JSON {kernel: {counter: mean per dispatch}}"""
import json
import sqlite3
import sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_to_json import _short


def main(paths):
    out = {}
    for path in paths:
        db = sqlite3.connect(path)
        cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {kcol}, counter_name, dispatch_id, sum(value) from counters_collection group by {kcol}, counter_name, dispatch_id"
        per = {}
        for k, c, _, v in db.execute(q):
            per.setdefault((_short(k), c), []).append(v)
        for (k, c), v in per.items():
            if "ctpvae" in k:
                out.setdefault(k, {})[c] = sum(v) / len(v)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:])
