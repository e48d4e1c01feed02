"""Compact event trace of a gfx950 .s file: barriers, MFMA runs, scratch traffic, vmcnt waits, LDS-DMA, branches.
usage: python tools/isa_events.py file.s [first_line last_line]"""
import re, sys
path = sys.argv[1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
pat = re.compile(r"^\s*(s_barrier|v_mfma\w*|scratch_\w+|s_waitcnt vmcnt\(\d+\)|global_load_lds\w*|global_load\w*|global_store\w*|s_cbranch\w*|s_endpgm|ds_read\w*|ds_write\w*|v_exp\w*|s_setprio \d)")
prev, cnt, first = None, 0, 0
def flush():
    if prev is not None:
        print(f"{first:6d} {prev}" + (f" x{cnt}" if cnt > 1 else ""))
for n, line in enumerate(open(path), 1):
    if n < lo or n > hi:
        continue
    if line.startswith(".LBB"):
        flush(); prev = None
        print(f"{n:6d} {line.split(':')[0]}:")
        continue
    m = pat.match(line)
    if not m:
        continue
    key = m.group(1)
    if key.startswith("s_cbranch"):
        key = " ".join(line.split()[:2])
    if key == prev:
        cnt += 1
    else:
        flush(); prev, cnt, first = key, 1, n
flush()
