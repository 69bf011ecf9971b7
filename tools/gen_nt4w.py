"""Generator of the hand-scheduled main loop of the 4-wave NT GEMM (tools/ubench_nt_tile.hip; round 5): emits tools/nt4w_asm.inc,
ONE inline-asm body for gfx950 -- 256 x 256 tile, 4 waves (2 x 2) of 128 x 128, one wave per SIMD, accumulators in a[0:255],
64-deep slabs, register-staged operand feed (16 x global_load_dwordx4 a slab ahead per wave, ds_write_b128 into the other LDS
buffer), fragments double-buffered over the four k-phases of a slab.

One slab = 64 MFMA gaps (phase kk = gap // 16, accumulator (i, j) = ((gap % 16) // 4, gap % 4)); every gap carries exactly ONE
memory instruction:
  even gaps : the 8 fragment reads of the NEXT phase, in the order the MFMAs need them (B0 A0 B1 B2 B3 A1 A2 A3); phase 3 reads
              (slab + 1, kk = 0) from the other LDS buffer, behind the slab's one s_barrier;
  odd gaps  : phases 0-2: ds_write_b128 of pieces 0-5 / 6-10 / 11-15 of slab + 1, then global_load_dwordx4 of slab + 2 into the
              registers just written (2 / 3 / 3 loads); phase 3: the other 8 loads.
Every s_waitcnt is COUNTED and computed here by walking the periodic instruction stream: LDS operations complete in order
(lgkmcnt), vector loads return in order (vmcnt); a consumer waits for exactly the operations issued up to its producer.
LDS: [A even 32 KB][A odd 32 KB][B even 32 KB][B odd 32 KB]; slab images and swizzle as in gemm_nt256_kernel (me_gemm.hip).
Fixed registers: FA v[16:47], FB v[48:79], R v[80:143], VOFF v[144:159], LWR v160, LRA v[162:169], LRB v[170:177]."""
import os

FA, FB, R, VOFF, LWR, LRA, LRB, TMP = 16, 48, 80, 144, 160, 162, 170, 178
SB, SCNT = 40, 44        # s[40:41] operand base pointer of this wave, s44 loop counter


def fa(buf, i): return "v[%d:%d]" % (FA + (buf * 4 + i) * 4, FA + (buf * 4 + i) * 4 + 3)
def fb(buf, j): return "v[%d:%d]" % (FB + (buf * 4 + j) * 4, FB + (buf * 4 + j) * 4 + 3)
def rr(p): return "v[%d:%d]" % (R + 4 * p, R + 4 * p + 3)
PROD = False            # production body (me_gemm_nt4w.inc): accumulators are "+a" operands %[c0] .. %[c15]


def acc(i, j):
    if PROD:
        return "%%[c%d]" % (i * 4 + j)
    return "a[%d:%d]" % ((i * 4 + j) * 16, (i * 4 + j) * 16 + 15)


READ_ORDER = [("B", 0), ("A", 0), ("B", 1), ("B", 2), ("B", 3), ("A", 1), ("A", 2), ("A", 3)]


def read_op(par, kk, kind, idx):
    buf = kk & 1
    if kind == "A":
        txt = "ds_read_b128 %s, v%d offset:%d" % (fa(buf, idx), LRA + 2 * kk + (idx & 1), idx * 4096 + par * 32768)
    else:
        txt = "ds_read_b128 %s, v%d offset:%d" % (fb(buf, idx), LRB + 2 * kk + (idx & 1), idx * 4096 + par * 32768)
    return {"k": "read", "frag": (kind, idx, buf), "txt": txt}


def slab_stream(par, no_loads=False):
    """instruction stream of one slab in LDS buffer par (dicts; waits are inserted later); no_loads: the gaps of the global
    loads stay empty (last slab of a tile: what it would fetch is never written to LDS)"""
    S = []
    writes = {0: [0, 1, 2, 3, 4, 5], 1: [6, 7, 8, 9, 10], 2: [11, 12, 13, 14, 15], 3: []}
    loads = {0: [0, 1], 1: [2, 3, 4], 2: [5, 6, 7], 3: [8, 9, 10, 11, 12, 13, 14, 15]}
    for kk in range(4):
        buf = kk & 1
        if kk == 3:
            S.append({"k": "barrier", "txt": "s_barrier"})
        even = [read_op(par, kk + 1, kd, ix) if kk < 3 else read_op(1 - par, 0, kd, ix) for kd, ix in READ_ORDER]
        odd = [{"k": "write", "p": p, "txt": "ds_write_b128 v%d, %s offset:%d" % (LWR, rr(p), p * 1024 + (1 - par) * 32768)} for p in writes[kk]]
        odd += [{"k": "nop" if no_loads else "load", "p": p,
                 "txt": "s_nop 0" if no_loads else "global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(p), VOFF + p, SB, SB + 1)} for p in loads[kk]]
        assert len(odd) == 8
        for g in range(16):
            i, j = g // 4, g % 4
            S.append({"k": "mfma", "needs": [("A", i, buf), ("B", j, buf)],
                      "txt": "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc(i, j), fb(buf, j), fa(buf, i), acc(i, j))})
            S.append(even[g // 2] if g % 2 == 0 else odd[g // 2])
    S.append({"k": "salu", "txt": "s_add_u32 s%d, s%d, 0x80" % (SB, SB)})
    S.append({"k": "salu", "txt": "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1)})
    return S


def insert_waits(body, prev=None):
    """body = one loop iteration; prev = the iteration executed before it (default: body itself, the periodic loop).  Returns the
    text lines with counted waits."""
    prev = body if prev is None else prev
    assert len(prev) == len(body)
    n = len(body)
    two = prev + body                       # iteration t - 1, iteration t
    is_lds = [op["k"] in ("read", "write") for op in two]
    is_vm = [op["k"] == "load" for op in two]
    out = []
    lds_done = -1                           # index (in `two`) up to which LDS operations are known complete
    vm_done = -1
    for x in range(n, 2 * n):
        op = two[x]
        need_lds, need_vm = -1, -1
        if op["k"] == "mfma":
            for fr in op["needs"]:
                y = max(k for k in range(x) if two[k]["k"] == "read" and two[k]["frag"] == fr)
                need_lds = max(need_lds, y)
        elif op["k"] == "write":
            y = max(k for k in range(x) if two[k]["k"] == "load" and two[k]["p"] == op["p"])
            need_vm = y
        elif op["k"] == "barrier":
            need_lds = max(k for k in range(x) if two[k]["k"] == "write")
        if need_lds > lds_done:
            cnt = sum(is_lds[need_lds + 1:x])
            if cnt <= 15:
                out.append("s_waitcnt lgkmcnt(%d)" % cnt)
            lds_done = need_lds             # (more than 15 younger operations: it has completed, the counter holds at most 15)
        if need_vm > vm_done:
            cnt = sum(is_vm[need_vm + 1:x])
            assert cnt <= 63
            out.append("s_waitcnt vmcnt(%d)" % cnt)
            vm_done = need_vm
        out.append(op["txt"])
    # the state carried into the next iteration is the same shifted by n: consistent because the stream is periodic
    return out


def setup(L):
    for p in range(16):
        L.append("v_mov_b32 v%d, %%[vo%d]" % (VOFF + p, p))
    L.append("v_mov_b32 v%d, %%[lwr]" % LWR)
    for kk in range(4):
        for q in range(2):
            L.append("v_xor_b32 v%d, 0x%x, %%[lra]" % (LRA + 2 * kk + q, (kk << 5) ^ (64 if q else 0)))
            L.append("v_xor_b32 v%d, 0x%x, %%[lrb]" % (LRB + 2 * kk + q, (kk << 5) ^ (64 if q else 0)))
    L.append("s_mov_b64 s[%d:%d], %%[sbase]" % (SB, SB + 1))
    L.append("s_mov_b32 s%d, %%[npair]" % SCNT)


def prologue(L):
    """slab 0 -> LDS buffer 0, slab 1 -> registers (the SAME issue order as in the loop: the loop's counted waits then hold from
    the first iteration), fragments of (0, kk = 0)"""
    for p in range(16):
        L.append("global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(p), VOFF + p, SB, SB + 1))
    L.append("s_add_u32 s%d, s%d, 0x80" % (SB, SB)); L.append("s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1))
    L.append("s_waitcnt vmcnt(0)")
    for p in range(16):
        L.append("ds_write_b128 v%d, %s offset:%d" % (LWR, rr(p), p * 1024))
    for p in range(16):
        L.append("global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(p), VOFF + p, SB, SB + 1))
    L.append("s_add_u32 s%d, s%d, 0x80" % (SB, SB)); L.append("s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 1))
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("s_barrier")
    for kd, ix in READ_ORDER:
        L.append(read_op(0, 0, kd, ix)["txt"])


def emit(path, macro, L, clob):
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_nt4w.py -- do not edit\n")
        f.write("#define %s_BODY \\\n" % macro)
        for ln in L:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("#define %s_CLOBBERS %s\n" % (macro, ", ".join('"%s"' % c for c in clob)))
    print("wrote", path, len(L), "instructions;", sum(1 for x in L if x.startswith("s_waitcnt")), "waits")


def main():
    global PROD
    here = os.path.dirname(os.path.abspath(__file__))
    body = None
    # ---- microbenchmark body (tools/ubench_nt_tile.hip): fixed a[0:255], reads two slabs past K, token write-out
    L = []
    setup(L)
    for n in range(256):
        L.append("v_accvgpr_write_b32 a%d, 0" % n)
    prologue(L)
    L.append("NT4W_LOOP_%=:")
    L.extend(insert_waits(slab_stream(0) + slab_stream(1)))
    L.append("s_sub_u32 s%d, s%d, 1" % (SCNT, SCNT))
    L.append("s_cmp_lg_u32 s%d, 0" % SCNT)
    L.append("s_cbranch_scc1 NT4W_LOOP_%=")
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    L.append("s_nop 15"); L.append("s_nop 15")
    L.append("v_mov_b32 %[osum], 0")
    for n in range(256):
        L.append("v_accvgpr_read_b32 v%d, a%d" % (TMP, n))
        L.append("s_nop 0")
        L.append("v_add_f32 %%[osum], %%[osum], v%d" % TMP)
    base_clob = ["v%d" % n for n in range(FA, TMP + 2)] + ["s%d" % n for n in range(SB, SCNT + 1)] + ["memory", "scc", "vcc"]
    emit(os.path.join(here, "nt4w_asm.inc"), "NT4W_ASM", L, base_clob + ["a%d" % n for n in range(256)])
    # ---- production body (midi-emotion_amd/csrc/me_gemm_nt4w.inc): one TILE.  Accumulators are operands (zero on entry, the C++
    #      write-out reads them).  The last slab pair is peeled: its first slab feeds from the tile's OWN first slab again (valid
    #      memory, L2-hot, never consumed) instead of reading past K, its second slab fetches nothing.  Ends with everything landed
    #      and the MFMA wait states served.
    #      (Measured and dropped, profiles/r05_nt_4wave.txt: feeding the NEXT tile's first slab from the peeled pair so that the next
    #      tile skips its load-wait-write prologue -- no gain: what the next tile waits for is not its prologue but the write-out's
    #      stores, which sit in front of its first loads in the in-order vmcnt queue.)
    PROD = True
    pair = insert_waits(slab_stream(0) + slab_stream(1))
    last_pair = insert_waits(slab_stream(0) + slab_stream(1, no_loads=True), prev=slab_stream(0) + slab_stream(1))
    L = []
    setup(L)
    prologue(L)
    L.append("s_cmp_le_u32 s%d, 1" % SCNT)
    L.append("s_cbranch_scc1 NT4W_LAST_%=")
    L.append("NT4W_LOOP_%=:")
    L.extend(pair)
    L.append("s_sub_u32 s%d, s%d, 1" % (SCNT, SCNT))
    L.append("s_cmp_gt_u32 s%d, 1" % SCNT)
    L.append("s_cbranch_scc1 NT4W_LOOP_%=")
    L.append("NT4W_LAST_%=:")
    L.append("s_mov_b64 s[%d:%d], %%[sbase]" % (SB, SB + 1))
    L.extend(last_pair)
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    L.append("s_nop 15")
    emit(os.path.join(here, "..", "midi-emotion_amd", "csrc", "me_gemm_nt4w.inc"), "ME_NT4W", L, base_clob)


if __name__ == "__main__":
    main()
