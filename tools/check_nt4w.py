"""(Kept for the record, see profiles/r06_nt_4wave.txt: needs the commit that holds gemm_nt4w_kernel.)
4-wave NT main loop (MIDIEMO_NT_MAINLOOP=3) against the 8-wave kernel: the switch is read when the library loads, so run this twice --
   MIDIEMO_NT_MAINLOOP=0 python tools/check_nt4w.py save   then   MIDIEMO_NT_MAINLOOP=3 python tools/check_nt4w.py compare
outputs of a set of shapes / write-outs must be bit-identical (same slab images, same k order per accumulator element)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
mode = sys.argv[1]
dts = {"bf16": torch.bfloat16, "fp16": torch.float16}
CASES = [(32768, 512, 2048), (32768, 512, 1536), (32768, 2048, 512), (32768, 1536, 512), (32768, 512, 512), (4096, 512, 256), (700, 1007, 512),
         (513, 300, 384), (256, 256, 256), (32768, 512, 1024)]
out = {}
for dn, dt in dts.items():
    for (M, N, K) in CASES:
        g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
        r = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(dt)
        A, B = r(M, K), r(N, K)
        bias, addt, gatet = torch.randn(N, device="cuda", generator=g), r(M, N), r(M, N)
        for name, kw in (("plain", {}), ("bias+relu", dict(bias=bias, flags=ops.ME_EPI_RELU)), ("add", dict(add=addt, bias=bias)),
                         ("gate", dict(gate=gatet, flags=ops.ME_EPI_RELU_BWD)), ("f32", dict(bias=bias, flags=ops.ME_EPI_OUT_F32))):
            C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if name == "f32" else dt)
            ops.gemm_nt(A, B, C, **kw)
            out[(dn, M, N, K, name)] = C.cpu()
torch.cuda.synchronize()
path = "/tmp/nt4w_ref.pt"
if mode == "save":
    torch.save(out, path); print("saved", len(out), "outputs")
else:
    ref = torch.load(path)
    bad = [k for k in out if not torch.equal(out[k].view(torch.int32 if out[k].dtype == torch.float32 else torch.int16), ref[k].view(torch.int32 if ref[k].dtype == torch.float32 else torch.int16))]
    print("compared", len(out), "outputs: differing", len(bad), bad[:6])
    for k in bad[:3]:
        d = (out[k].float() - ref[k].float()).abs()
        print(k, "max abs", float(d.max()), "nan", bool(torch.isnan(out[k].float()).any()))
