#!/usr/bin/env python3
"""Fails the build when, in a kernel whose name contains one of the given substrings, an instruction uses the result of a
matrix instruction (or a DPP instruction one of its source lanes' registers) sooner than gfx950 allows.

The matrix-core search reads its accumulators with vector instructions inside asm blocks.  The compiler pads its own
instructions against a matrix instruction's latency, but not the contents of an asm block, and the hardware does not
interlock: a reader that comes too early silently gets the old value (one wrong pixel in 10^5, found by test luck twice).
This walks the device assembly of the compilation (-save-temps) with a small issue model and checks every instruction:

  * a vector, LDS or memory instruction that reads or overwrites a register of a matrix instruction's result needs
    passes + 3 wait states between the two (8-pass fp4 32x32x64: 11);
  * a DPP instruction needs 2 wait states after a vector instruction that wrote the register it reads across lanes, and 5
    after a vector instruction's write of EXEC (v_cmpx).

Issue model (in wait states, i.e. issue slots of 4 clocks): every instruction takes one, `s_nop k` takes k + 1; a matrix
instruction issues when the matrix pipe is free and occupies it for its passes, so back-to-back products are `passes`
apart.  Everything else that delays a wavefront (operand waits, barriers, arbitration) only lengthens the distances, so the
model errs towards reporting.  Control flow: the scan follows the listing's order and, at every branch, also continues at
the branch target with the state of the branch (far enough for every pending result to have landed).

usage: check_mfma_hazards.py <listing.s> <substring>...     (exit status 1 and the offending lines on a violation)"""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOOKAHEAD = 64          # instructions followed at a branch target (the longest latency checked is 19 + 4 wait states)


def vregs(text):
    """The vector registers named in an operand, in order."""
    out = []
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def mfma_passes(mnemonic, rest):
    """Passes of a matrix instruction (its result lands passes + 3 wait states after the issue slot behind it)."""
    fp4_6 = {"2", "3", "4"}
    cbsz = re.search(r"cbsz:(\d+)", rest)
    blgp = re.search(r"blgp:(\d+)", rest)
    narrow = bool(cbsz and blgp and cbsz.group(1) in fp4_6 and blgp.group(1) in fp4_6)
    if "32x32x64_f8f6f4" in mnemonic:
        return 8 if narrow else 16
    if "16x16x128_f8f6f4" in mnemonic:
        return 4 if narrow else 8
    if "32x32" in mnemonic:
        return 16
    if "16x16" in mnemonic:
        return 8
    return 4


class Instr:
    __slots__ = ("line_no", "text", "mnemonic", "dst", "src", "is_mfma", "passes", "is_dpp", "dpp_src", "is_valu", "writes_exec",
                 "wait", "target", "is_branch", "uncond")

    def __init__(self, line_no, text):
        self.line_no, self.text = line_no, text
        parts = text.split(None, 1)
        self.mnemonic = m = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        ops = []
        depth, cur = 0, ""
        for ch in rest:                                  # operands are separated by commas outside brackets
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
        # modifiers ride on the last operand ("v3 row_ror:8 row_mask:0xf"): registers are named only in an operand's first token
        first_tok = [o.split()[0] if o.split() else "" for o in ops]
        self.is_mfma = m.startswith("v_mfma") or m.startswith("v_smfmac")
        self.passes = mfma_passes(m, rest) if self.is_mfma else 0
        self.is_valu = m.startswith("v_")
        self.is_dpp = m.endswith("_dpp") or " quad_perm:" in text or " row_" in text or " wave_" in text
        loads = m.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "buffer_load", "global_load", "flat_load", "scratch_load")) or "_rtn" in m
        no_dst = m.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop"))
        has_dst = bool(first_tok) and VREG.fullmatch(first_tok[0]) is not None and ((self.is_valu and not no_dst) or loads)
        self.dst = vregs(first_tok[0]) if has_dst else []
        self.src = [r for t in first_tok[(1 if has_dst else 0):] for r in vregs(t)]
        self.dpp_src = vregs(first_tok[1]) if self.is_dpp and has_dst and len(first_tok) > 1 else (self.src if self.is_dpp else [])
        # only a VECTOR instruction's write of EXEC (v_cmpx, or a v_cmp whose destination is exec) is a hazard for DPP; scalar writes are not
        self.writes_exec = m.startswith("v_cmpx") or (m.startswith("v_cmp") and bool(first_tok) and first_tok[0].startswith("exec"))
        k = re.match(r"s_nop\s+(\d+)", text)
        self.wait = int(k.group(1)) + 1 if k else 1
        self.is_branch = m.startswith(("s_cbranch", "s_branch"))
        self.uncond = m.startswith("s_branch")
        self.target = ops[0].split()[0] if self.is_branch and ops else None


def parse_functions(path, needles):
    """{name: (instructions, {label: index of the first instruction behind it})} of the listing's matching functions."""
    funcs, cur, name = {}, None, None
    for no, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.split(";", 1)[0].rstrip()
        m = re.match(r"^([\w.$]+):", line)
        if m:
            label = m.group(1)
            if not label.startswith(".L"):
                name = label if any(n in label for n in needles) else None
                cur = funcs.setdefault(name, ([], {})) if name else None
            elif cur is not None:
                cur[1][label] = len(cur[0])
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s:
            continue
        if s.startswith("."):
            if s.startswith(".size") or s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
                cur, name = None, None
            continue
        cur[0].append(Instr(no, s))
    return funcs


class State:
    """Issue time, when the matrix pipe is free, when each register of a matrix result lands, recent vector writes, last EXEC write."""

    def __init__(self):
        self.t, self.pipe_free = 0, 0
        self.ready, self.by = {}, {}             # register -> first slot a reader may issue in; the matrix instruction that wrote it
        self.valu_write, self.exec_write = {}, -100

    def copy(self):
        s = State()
        s.t, s.pipe_free, s.exec_write = self.t, self.pipe_free, self.exec_write
        s.ready, s.by, s.valu_write = dict(self.ready), dict(self.by), dict(self.valu_write)
        return s

    def pending(self):
        return any(v > self.t for v in self.ready.values())


def step(st, ins, fn, bad):
    if ins.is_mfma:
        issue = max(st.t, st.pipe_free)
        st.pipe_free = issue + ins.passes
    else:
        issue = st.t
        for r in set(ins.src) | set(ins.dst):
            if st.ready.get(r, 0) > issue:
                w = st.by[r]
                bad.add((fn, ins.line_no, "%s  touches v%d %d wait states too early after line %d: %s"
                         % (ins.text, r, st.ready[r] - issue, w.line_no, w.text)))
        if ins.is_dpp:
            for r in ins.dpp_src:
                if r in st.valu_write and issue - st.valu_write[r][0] - 1 < 2:
                    bad.add((fn, ins.line_no, "%s  reads v%d across lanes %d wait states after line %d wrote it (2 needed)"
                             % (ins.text, r, issue - st.valu_write[r][0] - 1, st.valu_write[r][1])))
            if issue - st.exec_write - 1 < 5:
                bad.add((fn, ins.line_no, "%s  DPP %d wait states after a vector write of EXEC (5 needed)" % (ins.text, issue - st.exec_write - 1)))
    if ins.is_mfma:
        for r in ins.dst:
            st.ready[r] = issue + ins.passes + 4             # passes + 3 wait states between the two instructions
            st.by[r] = ins
    elif ins.is_valu:
        for r in ins.dst:
            st.valu_write[r] = (issue, ins.line_no)
    if ins.writes_exec:
        st.exec_write = issue
    st.t = issue + ins.wait


def check_function(fn, instrs, labels, bad):
    st = State()
    for i, ins in enumerate(instrs):
        step(st, ins, fn, bad)
        if ins.is_branch and ins.target in labels:
            # the other way out of the branch: continue at its target with this state, as far as anything is pending
            side, j, n = st.copy(), labels[ins.target], 0
            while j < len(instrs) and n < LOOKAHEAD:
                step(side, instrs[j], fn, bad)
                if instrs[j].uncond and instrs[j].target in labels:
                    j = labels[instrs[j].target]
                else:
                    j += 1
                n += 1
        if ins.uncond:
            st = State()                                     # what follows is reached from elsewhere; those paths are walked from their branches
            st.t = st.pipe_free = 0


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    path, needles = sys.argv[1], sys.argv[2:]
    try:
        funcs = parse_functions(path, needles)
    except OSError as e:
        sys.exit("check_mfma_hazards: cannot read the listing: %s" % e)
    if not funcs:
        sys.exit("check_mfma_hazards: no function matching %s in %s" % (needles, path))
    bad, n_mfma = set(), 0
    for fn, (instrs, labels) in funcs.items():
        n_mfma += sum(1 for i in instrs if i.is_mfma)
        check_function(fn, instrs, labels, bad)
    if bad:
        lines = ["%s, line %d: %s" % b for b in sorted(bad, key=lambda b: (b[0], b[1]))]
        sys.exit("check_mfma_hazards: %d instruction(s) too close to what they depend on:\n  %s" % (len(lines), "\n  ".join(lines[:40])))
    print("check_mfma_hazards: %d kernels, %d matrix instructions, every reader of a result and every DPP source far enough" % (len(funcs), n_mfma))


if __name__ == "__main__":
    main()
