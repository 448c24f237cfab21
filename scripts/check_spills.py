#!/usr/bin/env python3
"""Scan gfx950 assembly of the kernel TU for a miscompilation seen with hipcc 7.2: register-allocator spill code
(scratch stores marked "Folded Spill", or v_accvgpr_write copies) placed at the top of a control-flow join block IN
FRONT OF the `s_or_b64 exec, exec, ...` that re-enables the lanes masked off by the branch.  Those lanes never store
their registers, and the reload after the join hands them whatever the scratch slot held (a 27-dim K = 25 kernel
produced MI = +-inf that way, and one wild gather faulted).  Usage: check_spills.py file.s [...]; exit status 1 if found.
Also prints, per filter kernel, the number of scratch spill instructions (0 is the goal for every shipped kernel)."""
import re
import sys


def scan(path):
    lines = open(path, errors="replace").read().split("\n")
    cur, bad, spills = None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1)
        if cur and "Folded Spill" in l:
            spills[cur] = spills.get(cur, 0) + 1
        if cur and l.startswith(".LBB"):
            j, n = i + 1, 0
            while j < len(lines):
                t = lines[j].strip()
                if not t or t.startswith(";"):
                    j += 1
                    continue
                if "Folded Spill" in t or t.startswith("v_accvgpr_write"):
                    n += 1
                    j += 1
                    continue
                if n and t.startswith("s_or_b64 exec, exec"):
                    bad.setdefault(cur, []).append(i + 1)
                break
    return bad, spills


def main():
    rc = 0
    for path in sys.argv[1:]:
        bad, spills = scan(path)
        for k, v in sorted(spills.items()):
            print("%s: %d scratch spill instructions in %s" % (path, v, k[:110]))
        for k, v in bad.items():
            rc = 1
            print("%s: SPILL CODE BEFORE EXEC RESTORE in %s at lines %s" % (path, k, v))
    return rc


if __name__ == "__main__":
    sys.exit(main())
