"""Instruction-class histogram of line ranges of a gfx950 .s file (the hot path of a loop body, cold blocks left out).
usage: python tools/isa_hist.py file.s  a-b [a-b ...]      classes: MFMA, VALU (incl. trans), LDS, VMEM, SALU, WAIT/NOP, BRANCH, BARRIER"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
cls = collections.Counter(); ops = collections.Counter()
for rng in sys.argv[2:]:
    a, b = map(int, rng.split("-"))
    for ln in lines[a - 1:b]:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): c = "MFMA"
        elif op.startswith(("ds_",)): c = "LDS"
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c = "VMEM"
        elif op.startswith(("s_waitcnt", "s_nop", "s_sleep")): c = "WAIT/NOP"
        elif op.startswith("s_barrier"): c = "BARRIER"
        elif op.startswith(("s_cbranch", "s_branch")): c = "BRANCH"
        elif op.startswith("s_"): c = "SALU"
        elif op.startswith("v_"): c = "VALU"
        else: c = "OTHER"
        cls[c] += 1; ops[re.sub(r"_e32|_e64|_sdwa|_dpp", "", op)] += 1
print(dict(cls), "total", sum(cls.values()))
for k, v in ops.most_common(60): print(f"  {v:4d} {k}")
