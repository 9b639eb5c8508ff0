#!/usr/bin/env python3
"""Condenses ThreadSanitizer reports of the emulated kernel to one line per racing pair of source lines (k_octree*.inc/.hip)."""
import collections
import re
import sys


def summarize(text):
    pairs = collections.Counter()
    for rep in text.split("WARNING: ThreadSanitizer: data race")[1:]:
        acc = re.findall(r"^\s+((?:Previous )?(?:[Aa]tomic )?(?:[Ww]rite|[Rr]ead)) of size (\d+) at \S+ by (?:thread T\d+|main thread).*?\n\s+#0 .*? (\S+?k_octree[^: ]*:\d+|\S+hip_runtime.h:\d+|\S+\.\w+:\d+) ", rep, re.M | re.S)
        locs = []
        for kind, size, where in acc[:2]:
            locs.append("%s(%s) %s" % (kind.replace("Previous ", "").lower(), size, where.split("/")[-1]))
        # frames below the shim's atomics / collectives: take the first k_octree frame of each stack
        stacks = re.split(r"\n\s*\n", rep)
        pairs[" <-> ".join(sorted(locs))] += 1
    return pairs


if __name__ == "__main__":
    for k, v in sorted(summarize(sys.stdin.read()).items(), key=lambda kv: -kv[1]):
        print("%6d  %s" % (v, k))
