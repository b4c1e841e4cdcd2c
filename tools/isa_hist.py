#!/usr/bin/env python3
"""Mnemonic histogram of one kernel in a hipcc -save-temps .s file (whole function; loops not unrolled
are counted once).  usage: isa_hist.py file.s substring-of-mangled-name"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
txt = open(path).read()
funcs = re.split(r"\n(?=\S+:\s*; @)", txt)
for f in funcs:
    head = f.split("\n", 1)[0]
    if key in head and "; @" in head:
        body = f.split("; -- End function")[0]
        h = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
                continue
            h[line.split()[0]] += 1
        tot = sum(h.values())
        f64 = sum(v for k, v in h.items() if k.endswith("_f64") or "_f64_" in k)
        print(head[:100])
        print("total %d   f64 %d   ds %d   global/buffer %d   s_waitcnt %d" % (
            tot, f64, sum(v for k, v in h.items() if k.startswith("ds_")),
            sum(v for k, v in h.items() if k.startswith(("global_", "buffer_", "flat_"))), h.get("s_waitcnt", 0)))
        for k, v in h.most_common(45):
            print("  %-28s %d" % (k, v))
