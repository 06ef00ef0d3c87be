#!/usr/bin/env python3
"""Fails the build when a kernel whose name contains one of the given substrings spills vector registers or touches
private (scratch) memory.

The matrix-core search reads its accumulators with DPP instructions inside asm blocks; a build whose registers spill
puts reloads between those instructions and has produced wrong disparities.  Input: the compiler's
-Rpass-analysis=kernel-resource-usage remarks and, with --asm, the device assembly of the same compilation: a kernel may
then reserve a private segment (the register scavenger's emergency slots of a build with many scalar spills) as long as no
instruction of it reads or writes private memory.   usage: check_no_scratch.py <remarks file> [--asm <listing>] <substring>..."""
import re
import sys


def scratch_instructions(asm_path, names):
    """Number of private-memory instructions per function of the listing."""
    counts, cur = {}, None
    for line in open(asm_path, errors="replace"):
        m = re.match(r"^(\w+):", line)
        if m:
            cur = m.group(1) if m.group(1) in names else (None if not line.startswith(".L") else cur)
            if cur:
                counts.setdefault(cur, 0)
            continue
        if cur and re.match(r"^\s+\.size\s", line):
            cur = None
        elif cur and re.match(r"^\s+(scratch_|buffer_(load|store)\w*\s.*\bs\[0:3\])", line):
            counts[cur] += 1                 # gfx950 reaches private memory with scratch_*; a buffer access through the private segment's descriptor counts as well
    return counts


def main():
    args = sys.argv[1:]
    asm = None
    if "--asm" in args:
        i = args.index("--asm")
        asm = args[i + 1]
        del args[i:i + 2]
    path, needles = args[0], args[1:]
    if asm is not None:
        try:
            open(asm).close()
        except OSError as e:
            sys.exit("check_no_scratch: the assembly listing given with --asm cannot be read: %s" % e)
    name, spills, reserved, seen = None, [], {}, 0
    for line in open(path, errors="replace"):
        m = re.search(r"remark:.* Function Name: (\S+)", line) if "Function Name" in line else None
        if m:
            name = m.group(1)
            if any(n in name for n in needles):
                seen += 1
            continue
        m = re.search(r"remark:.*\s(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line) if "remark" in line else None
        if m and name and any(n in name for n in needles) and int(m.group(2)) != 0:
            if m.group(1) == "VGPRs Spill":
                spills.append("%s: %s = %s" % (name, m.group(1), m.group(2)))
            else:
                reserved[name] = int(m.group(2))
    if seen == 0:
        sys.exit("check_no_scratch: no kernel matching %s in %s" % (needles, path))
    bad = list(spills)
    if reserved:
        counts = scratch_instructions(asm, set(reserved)) if asm else {}
        for k, size in reserved.items():
            if k not in counts:
                bad.append("%s: ScratchSize [bytes/lane] = %d (no assembly listing to look for its accesses in)" % (k, size))
            elif counts[k]:
                bad.append("%s: ScratchSize [bytes/lane] = %d, %d private-memory instructions" % (k, size, counts[k]))
    if bad:
        sys.exit("check_no_scratch: kernels that must stay in registers spill:\n  " + "\n  ".join(bad))
    note = "" if not reserved else " (%d reserve a private segment that no instruction touches)" % len(reserved)
    print("check_no_scratch: %d kernels, none spills or uses scratch%s" % (seen, note))


if __name__ == "__main__":
    main()
