"""generate.py -- emotion-conditioned MIDI token generation on the HIP engine.

Drop-in for the reference's src/generate.py: `generate()` keeps the reference
signature (generate.py:20-26) and sampling semantics (special-token exclusion,
note/rest temperatures, repeat penalty, top-k, top-p, multinomial; lines
122-189), and the CLI keeps the reference flags (lines 259-285).

What is new: the model call inside the loop is KV-cached (midiemo.decode) instead
of a full-window forward per token.  The cache is used while absolute positions
are stable; when the reference's sliding window starts moving (len > max_input_len)
or a per-step varying condition is given, the loop falls back to the reference's
full recompute, so token streams are identical to the reference in every case.
`--topk 1` is the greedy mode used for bit-exact parity.

Outputs per sample: inds_*.pt (token ids), txt_*.txt (symbols) and *.mid (Standard MIDI
File written by midiemo/midi_writer.py: the reference's tuples_to_mid semantics without the
pretty_midi dependency).
"""
import datetime
import os
import sys
from argparse import ArgumentParser
from copy import deepcopy

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "midi-emotion_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from midiemo import ops  # noqa: E402
from midiemo.decode import DecodeSession, WindowForward  # noqa: E402
from midiemo.models.build_model import build_model  # noqa: E402
from midiemo.midi_writer import write_midi  # noqa: E402
from midiemo.vocab import (emotion_symbols, get_maps, get_n_instruments, ind_list_to_str,  # noqa: E402
                           special_token_ids, timeshift_token_mask)


def sampling_temperature(prev_ids, repeat_counts, is_timeshift, temp_note, temp_rest, penalty_coeff):
    """Per-row temperature of one sampling step (generate.py:138-163): the note temperature right after a TIMESHIFT,
    else the rest temperature, raised by max(0, log((repeats + 1) / 4) * penalty_coeff) times itself."""
    dev = prev_ids.device
    temp = torch.where(is_timeshift[prev_ids], torch.tensor(float(temp_note), device=dev),
                       torch.tensor(float(temp_rest), device=dev))
    if penalty_coeff > 0:
        mult = torch.clamp(torch.log((repeat_counts + 1) / 4) * penalty_coeff, min=0)
        temp = temp + mult * temp
    return temp


def update_repeat_counts(repeat_counts, n_choices):
    """generate.py:186-189: a step with at most two choices counts as a repeat, any other halves the counter."""
    return torch.where(n_choices <= 2, repeat_counts + 1, torch.floor(repeat_counts / 2))


def generate(model, maps, device, out_dir, conditioning, short_filename=False,
             penalty_coeff=0.5, discrete_conditions=None, continuous_conditions=None,
             max_input_len=1024, amp=True, step=None,
             gen_len=2048, temperatures=[1.2, 1.2], top_k=-1,
             top_p=0.7, debug=False, varying_condition=None, seed=-1,
             verbose=False, primers=[["<START>"]], min_n_instruments=2,
             use_cache=True, return_ids=False, device_loop=True, use_window_graph=True):
    """Reference signature (generate.py:20-26) + `use_cache`, `return_ids`, `device_loop` (sampling loop replayed on
    the device as one HIP graph per token while the KV cache is valid; False = one Python iteration per token) and
    `use_window_graph` (once the window slides, its full forward is replayed as one HIP graph per token).
    `amp` is accepted for compatibility; the engine's precision is model.compute_dtype."""
    if not debug:
        os.makedirs(out_dir, exist_ok=True)
    model = model.to(device)
    model.eval()
    assert len(temperatures) in (1, 2)
    if len(temperatures) == 1:
        temperatures = [temperatures[0], temperatures[0]]

    discrete_conditions_tensor = None
    if varying_condition is not None:
        batch_size = varying_condition[0].size(0)
        continuous_conditions_t = None
    else:
        try:
            continuous_conditions_t = torch.tensor(continuous_conditions, dtype=torch.float32, device=device)
        except Exception:
            continuous_conditions_t = None
        if conditioning == "none":
            batch_size = len(primers)
        elif conditioning == "discrete_token":
            assert discrete_conditions is not None
            discrete_conditions_tensor = torch.tensor(
                [[maps["tuple2idx"][s] for s in sample] for sample in discrete_conditions],
                dtype=torch.long, device=device).t()                                   # [2, B]
            batch_size = discrete_conditions_tensor.size(1)
        else:
            batch_size = len(continuous_conditions)

    specials = torch.tensor(special_token_ids(maps), dtype=torch.int32, device=device)
    is_timeshift = torch.tensor(timeshift_token_mask(maps), dtype=torch.bool, device=device)

    if not isinstance(primers, list):
        primers = [[primers]]
    gen_inds = torch.tensor([[maps["tuple2idx"][s] for s in primer] for primer in primers], dtype=torch.long)
    null_cond = torch.full((batch_size, 2), float("nan"), device=device)
    if len(primers) == 1:
        gen_inds = gen_inds.repeat(batch_size, 1)

    if conditioning == "continuous_token":
        max_input_len -= 2                                        # generate.py:76
        conditions_tensor = continuous_conditions_t
    elif conditioning == "continuous_concat":
        conditions_tensor = continuous_conditions_t
    elif conditioning == "discrete_token":
        max_input_len -= discrete_conditions_tensor.size(0)       # generate.py:81
        conditions_tensor = null_cond
    else:
        conditions_tensor = null_cond
    if varying_condition is not None:
        varying_condition = [v.to(device) for v in varying_condition]

    gen_inds = gen_inds.t().contiguous().to(device)               # [P, B] time-major
    gen_song = torch.zeros((0, batch_size), dtype=torch.long, device=device)
    repeat_counts = torch.zeros(batch_size, device=device)
    temp_note, temp_rest = float(temperatures[0]), float(temperatures[1])
    V = model.vocab_size
    if V > 4096 and top_k != 1:
        # the device sampling tail (me_sample_topk_topp / me_sample_step) sorts one row of (value, id) pairs in LDS: 1024,
        # 2048 or 4096 wide (the reference's vocabularies have 1007 / 1017 / 1018 symbols).  There is no second, non-HIP
        # sampling path in this build; greedy decoding (--topk 1) has no limit.
        raise NotImplementedError("sampled generation supports vocabularies of at most 4096 symbols (got %d); "
                                  "use top_k=1 (greedy) or add a wider instantiation of sample_kernel" % V)

    cache_ok = bool(use_cache) and varying_condition is None
    sess = None
    win = None                                                    # WindowForward of the sliding regime (created on first use)
    fed = 0                                                      # tokens of gen_song already in the cache
    picked = torch.empty(batch_size, dtype=torch.long, device=device)
    n_choices_buf = torch.empty(batch_size, dtype=torch.int32, device=device)

    with torch.no_grad():
        i = 0
        while i < gen_len:
            i += 1
            gen_song = torch.cat((gen_song, gen_inds), 0)
            T = gen_song.size(0)
            if varying_condition is not None:
                conditions_tensor = torch.stack([varying_condition[0][:, i - 1], varying_condition[1][:, i - 1]], -1)

            if cache_ok and T <= max_input_len:
                # ---- incremental path: absolute positions are stable
                if (top_k != 1 and V <= 4096 and device_loop and sess is not None and fed == T - 1 and gen_len - i >= 2
                        and max_input_len - T >= 2):
                    # ---- the rest of the cache-valid span entirely on the device (DecodeSession.sample_run): the
                    # uniforms are drawn here, one torch.rand(batch_size) per step exactly like the eager loop below, so
                    # both paths consume the same random stream and produce the same tokens
                    n_dev = min(gen_len - i + 1, max_input_len - T + 1)
                    uni = torch.stack([torch.rand(batch_size, device=device) for _ in range(n_dev)])
                    ids_dev = sess.sample_run(gen_song[fed], n_dev, conditions_tensor, specials, is_timeshift, repeat_counts,
                                              temp_note, temp_rest, penalty_coeff, top_k, top_p, uni)      # [B, n_dev]
                    fed += n_dev
                    gen_song = torch.cat((gen_song, ids_dev[:, :n_dev - 1].t()), 0)
                    gen_inds = ids_dev[:, n_dev - 1].clone()[None, :]
                    i += n_dev - 1
                    continue
                if sess is None:
                    sess = DecodeSession(model, batch_size)
                    if conditioning == "continuous_token":
                        sess.prefill_condition_slots(conditions_tensor)
                    elif conditioning == "discrete_token":
                        for r in range(discrete_conditions_tensor.size(0)):
                            sess.step(discrete_conditions_tensor[r])
                while fed < T:
                    output = sess.step(gen_song[fed], conditions_tensor)
                    fed += 1
            else:
                # ---- reference path: full forward over the (sliding) window, generate.py:101-122
                input_ = gen_song[-max_input_len:] if T > max_input_len else gen_song
                if conditioning == "discrete_token":
                    input_ = torch.cat((discrete_conditions_tensor, input_), 0)
                if T > max_input_len and use_window_graph:
                    # the window has a fixed shape from here on: one captured HIP graph per token instead of ~55 eager launches
                    if win is None:
                        win = WindowForward(model)
                    output = win.last_logits(input_.t().contiguous(), conditions_tensor)
                else:
                    output = model(input_.t().contiguous(), conditions_tensor)[:, -1, :]

            if top_k == 1:
                # greedy: NaN->0, specials->-inf, argmax -- one kernel (generate.py:122-136,166-183)
                lg = output if output.is_contiguous() else output.contiguous()
                ops.greedy_pick(lg, V, specials, picked, batch_size)
                gen_inds = picked.clone()[None, :]
                repeat_counts += 1                                # one choice (<= 2) every step
                continue

            # ---- sampling tail (generate.py:122-189) in one launch: NaN->0, specials->-inf, log_softmax, per-row
            # temperature, top-k, nucleus cut, renormalise, draw, n_choices.  The draw is the inverse CDF at a uniform
            # from torch's generator (the reference's torch.multinomial stream itself is not reproducible).
            temp = sampling_temperature(gen_inds[0], repeat_counts, is_timeshift, temp_note, temp_rest, penalty_coeff)
            lg = output.float()
            lg = lg if lg.is_contiguous() else lg.contiguous()
            uni = torch.rand(batch_size, device=device)
            ops.sample_topk_topp(lg, V, specials, temp.float().contiguous(), top_k, top_p, uni, picked, n_choices_buf)
            gen_inds = picked.clone()[None, :]
            n_choices = n_choices_buf
            repeat_counts = update_repeat_counts(repeat_counts, n_choices)

    ids = gen_song.cpu()
    redo_primers, redo_discrete, redo_continuous = [], [], []
    for s in range(ids.size(1)):
        name = f"{s}" if short_filename else (
            (datetime.datetime.now().strftime("%Y_%m_%d_%H_%M_%S") if step is None else str(step)) + f"_{s}")
        if seed > 0:
            name += f"_s{seed}"
        if continuous_conditions_t is not None:
            c = [str(round(x, 2)).replace(".", "") for x in continuous_conditions_t[s].tolist()]
            name += f"_V{c[0]}_A{c[1]}"
        symbols = ind_list_to_str(ids[:, s].tolist(), maps)
        n_ins = get_n_instruments(symbols)
        if n_ins >= min_n_instruments:
            if not debug:
                torch.save(ids[:, s].clone(), os.path.join(out_dir, "inds_" + name + ".pt"))
                with open(os.path.join(out_dir, "txt_" + name + ".txt"), "w") as fh:
                    fh.write("\n".join(symbols))
                write_midi(os.path.join(out_dir, name + ".mid"), symbols)      # generate.py:216-232, no pretty_midi needed
                if verbose:
                    print(f"Saved to {os.path.join(out_dir, 'inds_' + name + '.pt')}")
        else:
            print(f"Only has {n_ins} instruments, not saving.")
            if conditioning == "none":
                redo_primers.append(primers[s])
                redo_discrete = None
                redo_continuous = None
            elif conditioning == "discrete_token":
                redo_discrete.append(discrete_conditions[s])
                redo_continuous = None
                redo_primers = primers
            else:
                redo_discrete = None
                redo_continuous.append(continuous_conditions_t[s].tolist())
                redo_primers = primers
    if return_ids:
        return ids
    return redo_primers, redo_discrete, redo_continuous


def main():
    parser = ArgumentParser()
    parser.add_argument('--model_dir', type=str, help='Directory with model', required=True)
    parser.add_argument('--no_cuda', action='store_true', help="(reference flag) the HIP engine has no CPU path")
    parser.add_argument('--num_runs', type=int, default=1)
    parser.add_argument('--gen_len', type=int, default=4096)
    parser.add_argument('--max_input_len', type=int, default=1216)
    parser.add_argument('--temp', type=float, nargs='+', default=[1.2, 1.2])
    parser.add_argument('--topk', type=int, default=-1)
    parser.add_argument('--topp', type=float, default=0.7)
    parser.add_argument('--debug', action='store_true')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--no_amp', action='store_true', help="run the exact-f32 tier instead of a 16-bit one")
    parser.add_argument('--compute_dtype', default=None, choices=["bf16", "fp16"],
                        help="16-bit tier (default: the one the checkpoint was trained in; fp16 = the reference's autocast dtype, generate.py:116)")
    parser.add_argument("--conditioning", type=str, required=True,
                        choices=["none", "discrete_token", "continuous_token", "continuous_concat"])
    parser.add_argument('--penalty_coeff', type=float, default=0.5)
    parser.add_argument("--quiet", action='store_true')
    parser.add_argument("--short_filename", action='store_true')
    parser.add_argument('--batch_size', type=int, default=4)
    parser.add_argument('--min_n_instruments', type=int, default=1)
    parser.add_argument('--valence', type=float, default=[None], nargs='+')
    parser.add_argument('--arousal', type=float, default=[None], nargs='+')
    parser.add_argument("--batch_gen_dir", type=str, default="")
    parser.add_argument("--output_root", type=str, default="../output", help="reference hard-codes ../output")
    parser.add_argument("--no_cache", action='store_true', help="reference-style full recompute every step")
    parser.add_argument("--no_device_loop", action='store_true', help="one Python iteration per token instead of the HIP-graph sampling loop")
    parser.add_argument("--no_window_graph", action='store_true', help="eager full forward per token in the sliding-window regime instead of the captured HIP graph")
    args = parser.parse_args()

    assert len(args.valence) == len(args.arousal), "Lengths of valence and arousal must be equal"
    assert (args.conditioning == "none") == (args.valence == [None] or args.arousal == [None]), \
        "If conditioning is used, specify valence and arousal; if not, don't"
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("generate.py: the MI355X engine needs a HIP device (no CPU fallback)")
    if args.seed > 0:
        torch.manual_seed(args.seed)
        torch.cuda.manual_seed(args.seed)

    model_root = os.path.join(args.output_root, args.model_dir)
    assert os.path.exists(model_root), model_root
    out_dir = os.path.join(model_root, "generations", "inference")
    if args.batch_gen_dir:
        out_dir = os.path.join(out_dir, args.batch_gen_dir)
    device = torch.device("cuda")

    mappings_fp = os.path.join(model_root, "mappings.pt")
    maps = torch.load(mappings_fp) if os.path.exists(mappings_fp) else get_maps(
        n_emotion_bins=5 if args.conditioning == "discrete_token" else 0)
    config = torch.load(os.path.join(model_root, "model_config.pt"))
    config["compute_dtype"] = "fp32" if args.no_amp else (args.compute_dtype or config.get("compute_dtype", "bf16"))
    model, _ = build_model(None, load_config_dict=config)
    model_fp = os.path.join(model_root, "model.pt")
    if not os.path.exists(model_fp):
        model_fp = os.path.join(model_root, "best_model.pt")
    model.load_state_dict(torch.load(model_fp, map_location="cpu"))
    model = model.to(device)

    n_bins = 5
    bins = np.linspace(-1 - 1e-12, 1 + 1e-12, num=n_bins + 1)
    v_syms, a_syms = emotion_symbols(n_bins, "V"), emotion_symbols(n_bins, "A")
    if args.valence == [None]:
        conditions = None
    elif len(args.valence) == 1:
        conditions = [[args.valence[0], args.arousal[0]] for _ in range(args.batch_size)]
    else:
        conditions = [[v, a] for v, a in zip(args.valence, args.arousal)]
    primers = [["<START>"]]
    discrete_conditions = None
    if args.conditioning == "discrete_token":
        discrete_conditions = [[v_syms[np.searchsorted(bins, v, side="right") - 1],
                                a_syms[np.searchsorted(bins, a, side="right") - 1]] for v, a in conditions]
    elif args.conditioning == "none":
        primers = [["<START>"] for _ in range(args.batch_size)]

    for _ in range(args.num_runs):
        p_run, d_run, c_run = deepcopy(primers), deepcopy(discrete_conditions), deepcopy(conditions)
        while not (p_run == [] or d_run == [] or c_run == []):
            p_run, d_run, c_run = generate(
                model, maps, device, out_dir, args.conditioning, discrete_conditions=d_run,
                min_n_instruments=args.min_n_instruments, continuous_conditions=c_run,
                penalty_coeff=args.penalty_coeff, short_filename=args.short_filename, top_p=args.topp,
                gen_len=args.gen_len, max_input_len=args.max_input_len, amp=not args.no_amp, primers=p_run,
                temperatures=args.temp, top_k=args.topk, debug=args.debug, verbose=not args.quiet, seed=args.seed,
                use_cache=not args.no_cache, device_loop=not args.no_device_loop,
                use_window_graph=not args.no_window_graph)


if __name__ == '__main__':
    main()
