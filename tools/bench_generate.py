"""Sampled generation throughput of generate() (headline model, B = 4, bf16): device-resident sampling loop vs one Python
iteration per token.  usage: python tools/bench_generate.py [gen_len]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd")); sys.path.insert(0, ROOT)
import torch
import generate as G
from midiemo.models.build_model import build_model
from midiemo.vocab import get_maps
gen_len = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(0)
model, _ = build_model(dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
                            conditioning="continuous_concat", dropout=0.1, compute_dtype="bf16"))
model = model.cuda().eval()
maps = get_maps()
conds = [[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]]
for dl in (True, False, True):
    torch.manual_seed(5)
    torch.cuda.synchronize(); t0 = time.time()
    ids = G.generate(model, maps, torch.device("cuda"), "/tmp/none", "continuous_concat", continuous_conditions=conds,
                     max_input_len=1024, gen_len=gen_len, debug=True, min_n_instruments=0, return_ids=True, device_loop=dl)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("device_loop=%-5s  %d x 4 tokens in %.3f s = %.0f tok/s   checksum %d" % (dl, gen_len, dt, 4 * gen_len / dt, int(ids.sum())))
