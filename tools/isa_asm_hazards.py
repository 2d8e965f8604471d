"""Checks of the inline-asm idioms of the K2 kernels that the compiler cannot check, on a hipcc -S listing:

1. Loads written as asm whose destination the compiler believes ready at the end of the asm statement (obs_load_ahead:
   `s_nop 4` + global_load_dwordx4; load_pair_entries: `s_nop 4` + two global_load_ushort): between such a load and the next
   s_waitcnt that names vmcnt, no instruction may touch its destination registers — a copy, spill or coalescing move the
   register allocator put there would read registers that are not yet written.  Whole functions are scanned:
   preheaders and exits, not just the loop bodies.
2. Every function that holds a `v_fma_f64 ... div:2` (the f64 output modifier of the Newton step, sfw_math.h::rsqrt_sqrt)
   switches the MODE register (s_setreg_imm32_b32 hwreg(HW_REG_MODE ...)) before its first such instruction.

usage: isa_asm_hazards.py <file.s>      prints one line per violation and a summary; exit status 1 if any."""
import re
import sys


def regs_of(tok):
    """VGPR numbers an operand token names: v7 -> {7}, v[4:7] -> {4,5,6,7}."""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def functions(lines):
    start = None
    for n, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l):
            start = n
        elif l.startswith(".Lfunc_end") and start is not None:
            yield lines[start].split(":")[0], start, n
            start = None


def main():
    lines = open(sys.argv[1]).read().split("\n")
    bad = n_loads = n_omod_fn = 0
    for name, a, b in functions(lines):
        body = lines[a:b]
        first_omod = next((n for n, l in enumerate(body) if "div:2" in l and "v_fma_f64" in l), None)
        if first_omod is not None:
            n_omod_fn += 1
            if not any("s_setreg_imm32_b32" in l and "HW_REG_MODE" in l for l in body[:first_omod]):
                print(f"{name}: v_fma_f64 div:2 at +{first_omod} without a MODE switch in front of it")
                bad += 1
        for n, l in enumerate(body):
            m = re.match(r"\s+global_load_(dwordx4|ushort)\s+(v\[\d+:\d+\]|v\d+),", l)
            if not m:
                continue
            # an asm-issued load: the s_nop 4 of the asm block stands right in front of it (or of its twin ushort load)
            if not any("s_nop 4" in x for x in body[max(0, n - 2):n]):
                continue
            n_loads += 1
            dst = regs_of(m.group(2))
            for k in range(n + 1, len(body)):
                x = body[k]
                if re.match(r"\s+s_waitcnt\b.*vmcnt", x):
                    break
                if not re.match(r"\s+[a-z]", x) or re.match(r"\s+(;|\.)", x):
                    continue
                ops = x.split(";")[0]
                if re.match(r"\s+global_load_ushort", ops) and "s_nop" not in ops and k == n + 1:
                    continue  # the twin load of load_pair_entries (its own destination)
                if regs_of(ops) & dst:
                    print(f"{name}: +{k} touches {sorted(regs_of(ops) & dst)} of the load at +{n} before its wait: {x.strip()}")
                    bad += 1
                    break
    print(f"{n_loads} asm-issued loads checked, {n_omod_fn} functions with an output-modifier fma, {bad} violation(s)")
    sys.exit(1 if bad else 0)


main()
