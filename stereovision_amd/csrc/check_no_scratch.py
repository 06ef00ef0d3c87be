#!/usr/bin/env python3
"""Fails the build when a kernel whose name contains one of the given substrings uses private (scratch) memory.

The matrix-core search reads its accumulators with DPP instructions inside asm blocks; a build whose registers spill
puts reloads between those instructions and has produced wrong disparities.  Input: the compiler's
-Rpass-analysis=kernel-resource-usage remarks.   usage: check_no_scratch.py <remarks file> <substring>..."""
import re
import sys


def main():
    path, needles = sys.argv[1], sys.argv[2:]
    name, bad, seen = None, [], 0
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            if any(n in name for n in needles):
                seen += 1
            continue
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
        if m and name and any(n in name for n in needles) and int(m.group(2)) != 0:
            bad.append("%s: %s = %s" % (name, m.group(1), m.group(2)))
    if seen == 0:
        sys.exit("check_no_scratch: no kernel matching %s in %s" % (needles, path))
    if bad:
        sys.exit("check_no_scratch: kernels that must stay in registers spill:\n  " + "\n  ".join(bad))
    print("check_no_scratch: %d kernels, none uses scratch" % seen)


if __name__ == "__main__":
    main()
