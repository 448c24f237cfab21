#!/usr/bin/env python3
"""Scan gfx950 assembly of the kernel TUs for a miscompilation seen with hipcc 7.2: register-allocator spill code placed
at the top of a control-flow join block IN FRONT OF the instruction that re-enables the lanes masked off by the branch.
Those lanes never store their registers, and the reload after the join hands them whatever the scratch slot held (a
27-dim K = 25 kernel produced MI = +-inf that way, and one wild gather faulted).

Rule (round 3: any EXEC-restore form, any spill form): walking from a block label (.LBB*) to the first instruction that can
ENABLE lanes (s_or_b64 / s_mov_b64 / s_xor_b64 with exec as destination, s_or_saveexec / s_xor_saveexec), flag any scratch
store or reload (lines the compiler marks "Folded Spill" / "Folded Reload") and any v_accvgpr_write / v_accvgpr_read copy
met on the way, with nothing but waits, nops, scalar moves and SGPR spill traffic in between (round 2's scanner wanted the
spill code directly behind the label and directly in front of `s_or_b64 exec, exec`).  The walk stops without a verdict at
the first instruction doing real work (a value parked or fetched after that is the block's own business: e.g. an `if` body
that reads a parked operand and ends in its own s_or), at a branch, at the next label, or at an instruction that only
narrows EXEC (s_and / s_andn2 forms, or s_mov from a pair and-ed in the block: the head of an inner `if`).

Usage: check_spills.py [--report FILE] file.s [...]; exit status 1 if the placement is found.  --report writes, per
kernel, VGPR / AGPR / SGPR counts, LDS and scratch bytes, the compiler's spill counts (from the .amdgpu_metadata block of
the same assembly) and the number of scratch spill instructions: the resource usage of the build that ships."""
import re
import sys

# instructions that can ENABLE lanes (a join or an else flip): or / mov / xor into exec, s_or_saveexec.  Narrowing forms
# (s_and_b64 exec, s_andn2_b64 exec, s_and_saveexec_b64: the head of an inner `if`) end the walk without a verdict: code in
# front of them ran for exactly the lanes that go on to use it.
EXEC_WIDEN = re.compile(r"^(s_or_b64|s_mov_b64|s_xor_b64|s_xnor_b64|s_orn2_b64|s_cselect_b64|s_or_saveexec_b64|s_xor_saveexec_b64)\s+(exec|s\[\d+:\d+\], *exec|s\[\d+:\d+\],)")
EXEC_NARROW = re.compile(r"^(s_and_b64|s_andn2_b64)\s+exec\b|^s_(and|andn2)_saveexec_b64\b")
SPILL = re.compile(r"Folded Spill|Folded Reload")
ACC = re.compile(r"^v_accvgpr_(write|read)")
# instructions that may sit between the spill code and the EXEC restore without making the block "real work": waits, nops,
# scalar moves / address arithmetic that do not write EXEC, SGPR spill traffic (v_readlane / v_writelane ignore EXEC)
NEUTRAL = re.compile(r"^(s_waitcnt|s_nop|s_mov_b32|s_mov_b64\s+(?!exec)|s_add_|s_addc_|s_lshl|s_load|v_readlane_b32|v_writelane_b32|s_barrier)")
TERMINATOR = re.compile(r"^(s_cbranch|s_branch|s_endpgm|s_setpc)")


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
            narrowed = set()  # SGPR pairs formed by s_and / s_andn2 inside this block: moving one into EXEC narrows it
            while j < len(lines):
                t = lines[j].strip()
                if not t or t.startswith(";"):
                    j += 1
                    continue
                if t.startswith(".LBB") or re.match(r"^_Z\S+:", t) or TERMINATOR.match(t):
                    break  # the block ends without touching EXEC: not a join with masked lanes
                m2 = re.match(r"^s_(and|andn2)_b64\s+(s\[\d+:\d+\]|vcc)\s*,", t)
                if m2:
                    narrowed.add(m2.group(2))
                m3 = re.match(r"^s_mov_b64\s+exec\s*,\s*(s\[\d+:\d+\]|vcc)", t)
                if EXEC_NARROW.match(t) or (m3 and m3.group(1) in narrowed):
                    break  # the head of an inner `if` (hipcc also writes it as s_mov s, exec; s_and t, s, cond; s_mov exec, t)
                if (EXEC_WIDEN.match(t) and re.match(r"^\S+\s+exec\b", t)) or re.match(r"^s_(or|xor)_saveexec_b64\b", t):
                    if n:
                        bad.setdefault(cur, []).append(i + 1)
                    break
                if SPILL.search(t) or ACC.match(t):
                    n += 1
                elif not NEUTRAL.match(t):
                    break  # real work starts before EXEC is touched: a value parked or fetched here is the block's own business
                j += 1
    return bad, spills


def metadata(path):
    """per-kernel resource figures from the .amdgpu_metadata block"""
    text = open(path, errors="replace").read()
    a, b = text.find(".amdgpu_metadata"), text.find(".end_amdgpu_metadata")
    out = {}
    if a < 0 or b < 0:
        return out
    for blk in re.split(r"\n  - \.agpr_count:", text[a:b])[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        out[g("name")] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                              scratch=g("private_segment_fixed_size"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                              wg=g("max_flat_workgroup_size"))
    return out


def demangle_short(name):
    m = re.search(r"(d19|d27)(\d+)(\w+?)(I.*)?E?v?NS_", name)
    ns = re.search(r"3rpf3(d19|d27)", name)
    k = re.search(r"kernelILi(\d+)ELb(\d)ELb(\d)ELi(\d)ELi(\d)E", name)
    base = re.search(r"\d+([a-z_]+kernel)", name)
    s = (ns.group(1) + "::" if ns else "") + (base.group(1) if base else name[:60])
    if k:
        s += "<K=%s,TL=%s,FAST=%s,NW=%s,PHASE=%s>" % k.groups()
    g = re.search(r"packed_kernelILi(\d+)E", name)
    if g:
        s += "<G=%s>" % g.group(1)
    return s


def main():
    args = sys.argv[1:]
    report = None
    if args and args[0] == "--report":
        report, args = args[1], args[2:]
    rc = 0
    rows = []
    for path in args:
        bad, spills = scan(path)
        meta = metadata(path)
        for k, v in sorted(spills.items()):
            print("%s: %d scratch spill instructions in %s" % (path, v, k[:110]))
        for k, v in bad.items():
            rc = 1
            print("%s: SPILL CODE BEFORE EXEC RESTORE in %s at lines %s" % (path, k, v))
        for k, m in sorted(meta.items()):
            rows.append("%-58s VGPR %3s AGPR %3s SGPR %3s  LDS(static) %6s B  scratch %5s B/lane  spilled VGPR %3s SGPR %3s  spill-store instrs %3d  %s"
                        % (demangle_short(k), m["vgpr"], m["agpr"], m["sgpr"], m["lds"], m["scratch"], m["vspill"], m["sspill"],
                           spills.get(k, 0), "HAZARD" if k in bad else "ok"))
    if report:
        with open(report, "w") as f:
            f.write("# resource usage of the shipped build, per kernel (scripts/check_spills.py --report, written by build.py)\n"
                    "# LDS is dynamic for the fused kernels (see DESIGN.md section 4); 'HAZARD' = spill code in front of an EXEC restore\n")
            f.write("\n".join(rows) + "\n")
    return rc


if __name__ == "__main__":
    sys.exit(main())
