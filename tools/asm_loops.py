"""Instruction mix of the loops of one kernel in a hipcc -S listing (development aid).
usage: python tools/asm_loops.py file.s <kernel-name-substring> [min_instr]"""
import re, sys, collections

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    min_instr = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and key in l and re.match(r"^[A-Za-z_]\w*:", l):
            start = i
        elif start is not None and l.strip().startswith(".amdhsa_kernel"):
            end = i; break
    body = lines[start:end]
    print(lines[start])
    labels, ins = {}, []
    for l in body:
        s = l.strip()
        m = re.match(r"^([.\w$]+):", s)
        if m:
            labels[m.group(1)] = len(ins); continue
        if not s or s.startswith((";", ".")): continue
        s = s.split(";")[0].strip()
        if s: ins.append(s)
    for idx, s in enumerate(ins):
        m = re.match(r"s_cbranch\w*\s+(\S+)|s_branch\s+(\S+)", s)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] >= min_instr:
                seg = ins[labels[tgt]:idx + 1]
                c = collections.Counter(classify(x.split()[0]) for x in seg)
                print("loop %s: %d instr  %s" % (tgt, len(seg), dict(c)))
                top = collections.Counter(x.split()[0] for x in seg).most_common(28)
                print("   ", ", ".join("%s:%d" % t for t in top))

main()
