"""HBM bytes of the FFN_pre forward / FFN_suf dX launches with the gate operand and with the ReLU sign mask (rocprofv3 PMC).
  python tools/pmc_relu_mask.py run       # the workload: 5 cases x 4 launches, in order (run under rocprofv3 --pmc ...)
  python tools/pmc_relu_mask.py report <fetch.db> <write.db>
Counter units and the gfx950 FETCH_SIZE doubling as in tools/hbm_traffic.py."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["fwd bias+relu (me_gemm_nt)", "fwd bias+relu + mask out", "bwd gate = activations (me_gemm_nt)", "bwd gate = sign mask", "plain product"]
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
    import torch
    from midiemo import ops
    dev, dt = "cuda", torch.bfloat16
    M, N, K = 32768, 2048, 512
    r = lambda *s: torch.randn(*s, device=dev).to(dt)
    A, W1, bias, dC, W2T = r(M, K), r(N, K), torch.randn(N, device=dev), r(M, K), r(N, K)
    hid, out = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt)
    mask = torch.zeros(ops.workspace_bytes(ops.ME_WS_RELU_MASK, M, N, K, dt), dtype=torch.uint8, device=dev)
    fns = [lambda: ops.gemm_nt(A, W1, hid, bias=bias, flags=ops.ME_EPI_RELU), lambda: ops.gemm_nt_relu_mask(A, W1, hid, mask, bias=bias),
           lambda: ops.gemm_nt(dC, W2T, out, gate=hid, flags=ops.ME_EPI_RELU_BWD), lambda: ops.gemm_nt_relu_mask(dC, W2T, out, mask, backward=True),
           lambda: ops.gemm_nt(dC, W2T, out)]
    torch.cuda.synchronize()
    for f in fns:
        for _ in range(4): f()
        torch.cuda.synchronize()
else:
    import sqlite3
    def seq(db, counter):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [x for x in tabs if x.startswith(p)][0]
        q = f"""select s.kernel_name, d.id, sum(e.value) from {t('rocpd_pmc_event_')} e join {t('rocpd_info_pmc_')} p on e.pmc_id = p.id
                join {t('rocpd_kernel_dispatch_')} d on d.event_id = e.event_id join {t('rocpd_info_kernel_symbol_')} s on s.id = d.kernel_id
                where p.name = '{counter}' group by d.id order by d.id"""
        return [(k, v) for k, _, v in c.execute(q) if "gemm_nt" in k]
    f, w = seq(sys.argv[2], "FETCH_SIZE"), seq(sys.argv[3], "WRITE_SIZE")
    assert len(f) == len(w) == 20, (len(f), len(w))
    print("M32768 N2048 K512 bf16, MB per launch (mean of launches 2-4 of each case; algorithmic: A 33.6 + W 2.1 + out 134.2 [+ gate 134.2 | mask 8.4])")
    for i, name in enumerate(CASES):
        rd = 2.0 * sum(v for _, v in f[4 * i + 1:4 * i + 4]) / 3 / 1e3
        wr = sum(v for _, v in w[4 * i + 1:4 * i + 4]) / 3 / 1e3
        print("  %-38s read %7.1f   write %7.1f" % (name, rd, wr))
