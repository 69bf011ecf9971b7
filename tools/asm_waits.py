"""Order of global memory ops, vmcnt waits, barriers and branches inside one loop of a kernel in a hipcc -S listing.
usage: python tools/asm_waits.py file.s <kernel-substring> <loop-label>"""
import re, sys
path, key, label = sys.argv[1:4]
lines = open(path).read().split("\n")
on = inl = False
n = 0
for l in lines:
    if not on and key in l and re.match(r"^[A-Za-z_]\w*:", l): on = True; continue
    if on and l.strip().startswith(".amdhsa_kernel"): break
    if on and l.startswith(label + ":"): inl = True; n = 0
    if inl:
        n += 1
        s = l.strip()
        if re.match(r"(global_|buffer_|s_waitcnt.*vmcnt|s_barrier|s_cbranch|s_branch|s_and_saveexec)", s):
            print("%4d  %s" % (n, s.split(";")[0][:80]))
        if re.match(r"s_cbranch\w*\s+" + re.escape(label) + r"\b", s) or re.match(r"s_branch\s+" + re.escape(label) + r"\b", s): break
