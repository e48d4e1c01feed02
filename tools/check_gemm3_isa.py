"""Build-time check of grouped_gemm_fp8_big3.hip's machine code (no GPU needed): (1) no compiler-generated instruction touches the fixed
fragment registers a140..a255; (2) no asm VMEM statement reads an SGPR that a VALU instruction (v_readlane / v_readfirstlane / v_cmp ..)
wrote fewer than 5 wait states earlier (hipcc's hazard recogniser does not look into asm statements).
usage: python tools/check_gemm3_isa.py [file.s]   (default: compiles the kernel to /tmp/big3_check.s; G3FLAGS adds compiler flags)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "sglang-fluentllm_amd/csrc/grouped_gemm_fp8_big3.hip")
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = "/tmp/big3_check.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-fno-slp-vectorize", "-Wno-inline-asm", "-Wno-unused-result", "-S", "--cuda-device-only", src, "-o", path]
                   + os.environ.get("G3FLAGS", "").split(), check=True, stderr=subprocess.DEVNULL)
lines = open(path).read().split("\n")
in_asm, bad_agpr, hazards = False, [], []
hist = []   # (text, in_asm) of the preceding instructions (straight-line approximation)


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


for n, ln in enumerate(lines, 1):
    t = ln.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    args = [a.strip() for a in t[len(op):].split(",")]
    if not in_asm:
        for a in re.findall(r"a\[(\d+):(\d+)\]|\ba(\d+)\b", t):
            hi = int(a[1]) if a[1] else int(a[2])
            if hi >= 140:
                bad_agpr.append((n, t))
    if in_asm and op.startswith(("global_load", "buffer_load")):
        used = set()
        for a in args:
            if a:
                used |= sregs(a.split()[0])
        states = 0
        for pt, pasm in reversed(hist):
            pop = pt.split()[0]
            if pop.startswith("v_") and not pasm:
                dst = pt[len(pop):].split(",")[0].strip()
                if sregs(dst) & used and states < 5:
                    hazards.append((n, t, pt, states))
            states += (int(pt.split()[1]) + 1) if pop == "s_nop" else 1
            if states >= 5:
                break
    hist.append((t, in_asm))
    hist = hist[-12:]
print(f"{path}: compiler instructions touching a140+: {len(bad_agpr)}; unpadded VALU->SGPR->VMEM hazards: {len(hazards)}")
for x in bad_agpr[:10]:
    print("  AGPR", x)
for x in hazards[:10]:
    print("  HAZARD", x)
sys.exit(1 if bad_agpr or hazards else 0)
