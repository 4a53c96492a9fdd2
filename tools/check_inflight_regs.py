"""Build-time guard for the hand-managed epilogues (gemm_common.h e4_*): no instruction may touch a register that an inline-asm global load
has in flight.  The residual look-ahead loads are asm (the compiler does not know their destinations are pending) and their waits are counted
asm s_waitcnt vmcnt(N) that NAME the registers they cover; everything is correct as long as the register allocator leaves those registers alone
between the load and its wait.  Passing the pieces by value through the unrolled fragments can make it insert a v_mov of such a register
(seen with E4_DEPTH_LINEAR=3 once the compiler's own vmcnt(0) drains were gone: wrong results, profiles/r06_e4_asm_reads_ab.txt) - this script
finds that in the disassembly.  Scan per kernel in program order: VMEM operations are numbered as they appear and an s_waitcnt vmcnt(N) retires all
but the N youngest; forward branches carry the set of pending registers to their labels (the code behind an unconditional branch starts from what
its own label receives), backward branches are ignored (the main loops hold no asm loads).
Known false positive (product build): the two multi-tap <.., 1, .., true> GroupNorm-statistics kernels of gemm.hip are reported through a path that takes the
`no residual operand` branch around a counted wait AFTER having issued residual loads (the test is launch-invariant; the scanner only recognises
the simple forms of that correlation).  What a real finding looks like: v_mov_b64 copies whose SOURCE is pending (EXTRA_FLAGS="-DE4_DEPTH_LINEAR=3").
usage: python tools/check_inflight_regs.py v3d_amd/csrc/gemm.hip [kernel-name-substring ...]      (--strict as first argument: exit code 1 on a finding)
       EXTRA_FLAGS="-DE4_DEPTH_LINEAR=3" python tools/check_inflight_regs.py v3d_amd/csrc/gemm.hip          audit an A/B build"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffast-math", "-fno-finite-math-only"]
VMEM = ("global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "flat_load", "flat_store", "global_atomic", "buffer_atomic")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(name, body):
    """A path state = the VMEM operations still outstanding on that path, oldest first, each with the registers an ASM load among them will write.
    Paths that meet at a label are kept side by side (a counted wait counts from the young end: two paths with a different number of stores between
    the load and the wait cannot be merged position by position); identical states collapse."""
    findings = []
    states = {()}            # set of tuples of frozensets
    in_asm = False
    at_label = {}            # label -> states carried by the forward branches to it
    reachable = True
    seen = set()
    lines = body.split("\n")
    label_at = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"^(\.LBB[0-9_]+):", l.strip())] if m}

    def skips_residual_loads(ln, lab):
        """a conditional branch around asm loads (or their counted wait) that are not followed by a full drain = the `no residual operand` test, the same for every
        fragment of a launch: the path that skips the loads never has loads pending, so it carries nothing (correlated branches, not merged)"""
        k1 = label_at.get(lab, -1)
        if k1 <= ln:
            return False
        region = "\n".join(lines[ln:k1])
        return ("ASMSTART\n\tglobal_load" in region or "ASMSTART\n\ts_waitcnt vmcnt(" in region) and "vmcnt(0)" not in region

    def cap(st):
        if len(st) <= 64:
            return st
        return set(sorted(st, key=lambda t: (-sum(len(x) for x in t), len(t)))[:64])      # keep the states with the most pending registers

    for ln, raw in enumerate(lines):
        line = raw.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            lab = m.group(1)
            seen.add(lab)
            carried = at_label.pop(lab, None)
            if carried is not None:
                states = cap(states | carried) if reachable else set(carried)
                reachable = True
            elif not reachable:
                states = {()}    # (reached by backward branches only: a loop header behind an unconditional branch)
                reachable = True
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        line = line.split(";")[0].strip()
        mnem, _, rest = line.partition(" ")
        if mnem in ("s_cbranch_execz", "s_cbranch_execnz"):
            continue             # the compiler's skip over a region no lane executes: the counted waits are written for the path through it
        if mnem == "s_branch" or mnem.startswith("s_cbranch"):
            lab = rest.strip()
            if lab not in seen and reachable and not (mnem != "s_branch" and skips_residual_loads(ln, lab)):
                at_label[lab] = cap(at_label.get(lab, set()) | states)
            if mnem == "s_branch":
                reachable = False
                states = {()}
            continue
        if not reachable:
            continue
        if mnem == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                states = {(t[len(t) - n:] if n < len(t) else t) if n else () for t in states}
            continue
        ops = [o.strip() for o in rest.split(",")] if rest else []
        pend = set()
        for t in states:
            for x in t:
                pend |= x
        touched = set()
        for o in ops:
            touched |= regs(o)
        bad = touched & pend
        if bad:
            findings.append((ln, line, sorted(bad)))
        if mnem.startswith(VMEM):
            is_load = "_load" in mnem and " lds" not in (" " + rest)
            new = frozenset(regs(ops[0])) if is_load and in_asm and ops else frozenset()
            states = {(t + (new,))[-64:] for t in states}
    return findings


def main():
    args = sys.argv[1:]
    strict = bool(args) and args[0] == "--strict"
    if strict:
        args = args[1:]
    src, pats = args[0], args[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        extra = ["-fno-slp-vectorize"] if src.endswith(("ff.hip", "attn.hip")) else []
        extra += os.environ.get("EXTRA_FLAGS", "").split()
        subprocess.run(["hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    parts = re.split(r"\n(_Z[^\n:]+):[^\n]*\n", text)
    dirty = 0
    nk = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if "global_load" not in body or "ASMSTART" not in body or (pats and not any(p in name for p in pats)):
            continue
        body = body.split(".section")[0]
        nk += 1
        f = scan(name, body)
        if f:
            dirty += 1
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"DIRTY {dem[:140]}: {len(f)} instruction(s) touch a register with an asm load in flight")
            for ln, line, bad in f[:6]:
                print(f"      line {ln}: {line}    <- v{bad}")
    print(f"{src}: {nk} kernels with asm loads scanned, {dirty} dirty")
    sys.exit(1 if dirty and strict else 0)


if __name__ == "__main__":
    main()
