"""(Kept for the record: the kernel it feeds, gemm_nt4w_kernel, is NOT in the library -- profiles/r06_nt_4wave.txt; `git log -S gemm_nt4w_kernel`
finds the commit that holds it.)
Generator of the hand-scheduled main loop of the 4-wave NT GEMM (round 6: round 5's loop, deleted with the kernel it belonged to,
with a SECOND register stage in the operand feed -- VERDICT r5 item 4): emits midi-emotion_amd/csrc/me_gemm_nt4w.inc, ONE inline-asm
body for gfx950 -- 256 x 256 tile, 4 waves (2 x 2) of 128 x 128, one wave per SIMD, the 256 accumulators as "+a" operands,
64-deep slabs, fragments double-buffered over the four k-phases of a slab.

Operand feed.  A wave stages 16 pieces (8 rows x 128 B each) of its operand per slab: global_load_dwordx4 into registers,
ds_write_b128 into the other LDS buffer one slab before the MFMAs read it.  Round 5's loop had ONE register stage: a load issued
during slab t was written during slab t + 1 -- 1.3 us of latency cover, enough for L2 / Infinity-Cache-warm operands (1 110 TF/s in
a replay) and not for the step's cold activation operand (68.4 -> 81.4 us in the live step, profiles/r05_nt_4wave.txt).  With one
wave per SIMD the register file has room for a second stage: two sets of 64 registers alternate by slab parity, slab t writes the
set loaded during slab t - 2 and requests slab t + 3 into it: 32 loads in flight per wave, 2.6 us of cover.

One slab = 64 MFMA gaps (phase kk = gap // 16, accumulator (i, j) = ((gap % 16) // 4, gap % 4)); every gap carries exactly ONE
memory instruction:
  even gaps : the 8 fragment reads of the NEXT phase, in the order the MFMAs need them (B0 A0 B1 B2 B3 A1 A2 A3); phase 3 reads
              (slab + 1, kk = 0) from the other LDS buffer, behind the slab's one s_barrier;
  odd gaps  : phases 0-2: ds_write_b128 of pieces 0-5 / 6-10 / 11-15 of slab + 1, then global_load_dwordx4 of slab + 3 into the
              registers just written (2 / 3 / 3 loads); phase 3: the other 8 loads.
Every s_waitcnt is COUNTED and computed here by walking the periodic instruction stream: LDS operations complete in order
(lgkmcnt), vector loads return in order (vmcnt); a consumer waits for exactly the operations issued up to its producer.
The end of a tile is peeled: the last THREE slabs request nothing (what they would fetch lies behind the contraction range).
LDS: [A even 32 KB][A odd 32 KB][B even 32 KB][B odd 32 KB]; slab images and swizzle as in gemm_nt256_kernel (me_gemm.hip).
Feed split (second version): every wave stages 8 pieces of A and 8 of B (the first version gave waves 0 / 1 the sixteen A pieces and
waves 2 / 3 the B pieces: all of the step's cold loads sat in two waves).
Fixed registers: FA v[16:47], FB v[48:79], R0 v[80:143], VOFF v[144:159], LWR v160 (A) / v161 (B), LRA v[162:169], LRB v[170:177],
R1 v[180:243]."""
import os

FA, FB, R0, VOFF, LWR, LWRB, LRA, LRB, R1 = 16, 48, 80, 144, 160, 161, 162, 170, 180
LAST_V = R1 + 63
SB, SB2, SCNT = 40, 42, 44        # s[40:41] / s[42:43] A / B base pointers of the tile, s44 loop counter


def lw(p): return LWR if p < 8 else LWRB
def sb(p): return SB if p < 8 else SB2


def fa(buf, i): return "v[%d:%d]" % (FA + (buf * 4 + i) * 4, FA + (buf * 4 + i) * 4 + 3)
def fb(buf, j): return "v[%d:%d]" % (FB + (buf * 4 + j) * 4, FB + (buf * 4 + j) * 4 + 3)
def rr(st, p): return "v[%d:%d]" % ((R1 if st else R0) + 4 * p, (R1 if st else R0) + 4 * p + 3)
def acc(i, j): return "%%[c%d]" % (i * 4 + j)


READ_ORDER = [("B", 0), ("A", 0), ("B", 1), ("B", 2), ("B", 3), ("A", 1), ("A", 2), ("A", 3)]


def read_op(par, kk, kind, idx):
    buf = kk & 1
    if kind == "A":
        txt = "ds_read_b128 %s, v%d offset:%d" % (fa(buf, idx), LRA + 2 * kk + (idx & 1), idx * 4096 + par * 32768)
    else:
        txt = "ds_read_b128 %s, v%d offset:%d" % (fb(buf, idx), LRB + 2 * kk + (idx & 1), idx * 4096 + par * 32768)
    return {"k": "read", "frag": (kind, idx, buf), "txt": txt}


def slab_stream(par, mfma, no_loads=False):
    """instruction stream of one slab in LDS buffer par (dicts; waits are inserted later).  Slab t (t & 1 = par) writes slab t + 1
    from register set 1 - par (loaded two slabs ago) and requests slab t + 3 into it.  no_loads: the load gaps stay empty."""
    S = []
    st = 1 - par
    writes = {0: [0, 1, 2, 3, 4, 5], 1: [6, 7, 8, 9, 10], 2: [11, 12, 13, 14, 15], 3: []}
    loads = {0: [0, 1], 1: [2, 3, 4], 2: [5, 6, 7], 3: [8, 9, 10, 11, 12, 13, 14, 15]}
    for kk in range(4):
        buf = kk & 1
        if kk == 3:
            S.append({"k": "barrier", "txt": "s_barrier"})
        even = [read_op(par, kk + 1, kd, ix) if kk < 3 else read_op(1 - par, 0, kd, ix) for kd, ix in READ_ORDER]
        odd = [{"k": "write", "p": (st, p), "txt": "ds_write_b128 v%d, %s offset:%d" % (lw(p), rr(st, p), (p % 8) * 1024 + (1 - par) * 32768)}
               for p in writes[kk]]
        odd += [{"k": "nop" if no_loads else "load", "p": (st, p),
                 "txt": "s_nop 0" if no_loads else "global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(st, p), VOFF + p, sb(p), sb(p) + 1)} for p in loads[kk]]
        assert len(odd) == 8
        for g in range(16):
            i, j = g // 4, g % 4
            S.append({"k": "mfma", "needs": [("A", i, buf), ("B", j, buf)],
                      "txt": "%s %s, %s, %s, %s" % (mfma, acc(i, j), fb(buf, j), fa(buf, i), acc(i, j))})
            S.append(even[g // 2] if g % 2 == 0 else odd[g // 2])
    for b in (SB, SB2):
        S.append({"k": "salu", "txt": "s_add_u32 s%d, s%d, 0x80" % (b, b)})
        S.append({"k": "salu", "txt": "s_addc_u32 s%d, s%d, 0" % (b + 1, b + 1)})
    return S


def insert_waits(body, prev=None):
    """body = one loop iteration (a slab pair); prev = the iteration executed before it (default: body itself, the periodic loop).
    Returns the text lines with counted waits."""
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
            ys = [k for k in range(x) if two[k]["k"] == "load" and two[k]["p"] == op["p"]]
            if ys:                          # (none: the producer was peeled away -- the piece is never consumed)
                need_vm = max(ys)
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
    return out


def setup(L):
    # the sixteen piece offsets of the lane come through the wave's (idle) write-out staging area: as sixteen "v" inputs they did
    # not fit beside the 228 registers the body claims (inputs cannot live in clobbered registers)
    for q in range(4):
        L.append("ds_read_b128 v[%d:%d], %%[vstg] offset:%d" % (VOFF + 4 * q, VOFF + 4 * q + 3, q * 1024))
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("v_mov_b32 v%d, %%[lwr]" % LWR)
    L.append("v_add_u32 v%d, 0x10000, %%[lwr]" % LWRB)
    for kk in range(4):
        for q in range(2):
            L.append("v_xor_b32 v%d, 0x%x, %%[lra]" % (LRA + 2 * kk + q, (kk << 5) ^ (64 if q else 0)))
            L.append("v_xor_b32 v%d, 0x%x, %%[lrb]" % (LRB + 2 * kk + q, (kk << 5) ^ (64 if q else 0)))
    L.append("s_mov_b64 s[%d:%d], %%[sbase]" % (SB, SB + 1))
    L.append("s_mov_b64 s[%d:%d], %%[sbaseb]" % (SB2, SB2 + 1))
    L.append("s_mov_b32 s%d, %%[nloop]" % SCNT)


def adv(L):
    for b in (SB, SB2):
        L.append("s_add_u32 s%d, s%d, 0x80" % (b, b)); L.append("s_addc_u32 s%d, s%d, 0" % (b + 1, b + 1))


def prologue(L):
    """slab 0 -> set 0 -> LDS buffer 0; slab 1 -> set 1, slab 2 -> set 0 stay in flight in the loop's own issue order (so the
    loop's counted waits hold from its first iteration); fragments of (0, kk = 0)"""
    for p in range(16):
        L.append("global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(0, p), VOFF + p, sb(p), sb(p) + 1))
    adv(L)
    for p in range(16):
        L.append("global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(1, p), VOFF + p, sb(p), sb(p) + 1))
    adv(L)
    L.append("s_waitcnt vmcnt(16)")
    for p in range(16):
        L.append("ds_write_b128 v%d, %s offset:%d" % (lw(p), rr(0, p), (p % 8) * 1024))
    for p in range(16):
        L.append("global_load_dwordx4 %s, v%d, s[%d:%d]" % (rr(0, p), VOFF + p, sb(p), sb(p) + 1))
    adv(L)
    L.append("s_waitcnt lgkmcnt(0)")
    L.append("s_barrier")
    for kd, ix in READ_ORDER:
        L.append(read_op(0, 0, kd, ix)["txt"])


def emit(path, macro, L, clob):
    with open(path, "a") as f:
        f.write("#define %s_BODY \\\n" % macro)
        for ln in L:
            f.write('    "%s\\n\\t" \\\n' % ln)
        f.write('    ""\n')
        f.write("#define %s_CLOBBERS %s\n" % (macro, ", ".join('"%s"' % c for c in clob)))
    print("wrote", path, macro, len(L), "instructions;", sum(1 for x in L if x.startswith("s_waitcnt")), "waits")


def body(mfma):
    """one TILE (K >= 256, K % 128 == 0): prologue; (K / 128 - 2) x the periodic slab pair; the pair whose second slab requests
    nothing; the pair that requests nothing.  Accumulators are operands (zero on entry, the C++ write-out reads them).  Ends with
    everything landed and the MFMA wait states served."""
    pair = slab_stream(0, mfma) + slab_stream(1, mfma)
    pair2 = slab_stream(0, mfma) + slab_stream(1, mfma, no_loads=True)
    last = slab_stream(0, mfma, no_loads=True) + slab_stream(1, mfma, no_loads=True)
    L = []
    setup(L)
    prologue(L)
    L.append("s_cmp_eq_u32 s%d, 0" % SCNT)
    L.append("s_cbranch_scc1 NT4W_TAIL_%=")
    L.append("NT4W_LOOP_%=:")
    L.extend(insert_waits(pair))
    L.append("s_sub_u32 s%d, s%d, 1" % (SCNT, SCNT))
    L.append("s_cmp_lg_u32 s%d, 0" % SCNT)
    L.append("s_cbranch_scc1 NT4W_LOOP_%=")
    L.append("NT4W_TAIL_%=:")
    L.extend(insert_waits(pair2, prev=pair))
    L.extend(insert_waits(last, prev=pair2))
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    L.append("s_nop 15")
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "midi-emotion_amd", "csrc", "me_gemm_nt4w.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_nt4w.py -- do not edit\n")
    clob = ["v%d" % n for n in range(FA, LAST_V + 1)] + ["s%d" % n for n in range(SB, SCNT + 1)] + ["memory", "scc", "vcc"]
    emit(path, "ME_NT4W_BF16", body("v_mfma_f32_32x32x16_bf16"), clob)
    emit(path, "ME_NT4W_F16", body("v_mfma_f32_32x32x16_f16"), clob)


if __name__ == "__main__":
    main()
