"""Token vocabulary of the reference (data/data_processing.py:183-222), restated so that
generate.py can map special symbols and count instruments without the dataset tooling.

1007 base tokens: <PAD>=0, <START>=1, then for each of 5 instruments x {OFF, ON} the 88
pitches 21..108, then 125 TIMESHIFT steps of 8 ms (8..1000).  `discrete_token` models append
the sorted emotion-bin symbols (data/loader.py:58-75) -> 1017 tokens."""

INSTRUMENTS = ["DRUMS", "GUITAR", "BASS", "PIANO", "STRINGS"]


def get_maps(min_pitch=21, max_pitch=108, max_timeshift=1000, timeshift_step=8, n_emotion_bins=0):
    token_syms = ["<PAD>", "<START>"]
    event_syms = []
    for ins in INSTRUMENTS:
        for on_off in ("OFF", "ON"):
            name = "%s_%s" % (on_off, ins)
            event_syms.append(name)
            token_syms += [(name, pitch) for pitch in range(min_pitch, max_pitch + 1)]
    token_syms += [("TIMESHIFT", ts) for ts in range(timeshift_step, max_timeshift + timeshift_step, timeshift_step)]
    event_syms.append("TIMESHIFT")
    maps = {"event2idx": {s: i for i, s in enumerate(event_syms)}, "idx2event": dict(enumerate(event_syms))}
    maps["tuple2idx"], maps["idx2tuple"] = {}, {}
    for idx, sym in enumerate(token_syms):
        key = (maps["event2idx"][sym[0]], sym[1]) if isinstance(sym, tuple) else sym
        maps["tuple2idx"][key] = idx
        maps["idx2tuple"][idx] = key
    if n_emotion_bins:
        extra = sorted(emotion_symbols(n_emotion_bins, "V") + emotion_symbols(n_emotion_bins, "A"))
        for sym in extra:
            idx = len(maps["idx2tuple"])
            maps["tuple2idx"][sym] = idx
            maps["idx2tuple"][idx] = sym
    return maps


def emotion_symbols(n_bins, letter):
    """generate.py:320-328 / data/preprocess_features.py:48-53."""
    if n_bins % 2 == 0:
        ids = list(range(-n_bins // 2, 0)) + list(range(1, n_bins // 2 + 1))
    else:
        ids = list(range(-(n_bins - 1) // 2, (n_bins - 1) // 2 + 1))
    return ["<%s%d>" % (letter, i) for i in ids]


def special_token_ids(maps):
    """Indices whose symbol starts with '<' (generate.py:57)."""
    return sorted(i for i, s in maps["idx2tuple"].items() if isinstance(s, str) and s[0] == "<")


def timeshift_token_mask(maps):
    """bool list over the vocabulary: token is a TIMESHIFT event (generate.py:138-150)."""
    ts = maps["event2idx"]["TIMESHIFT"]
    return [isinstance(s, tuple) and s[0] == ts for _, s in sorted(maps["idx2tuple"].items())]


def ind_list_to_str(ids, maps):
    """data/data_processing_reverse.py:55-80."""
    out = []
    for i in ids:
        s = maps["idx2tuple"][int(i)]
        out.append(s if isinstance(s, str) else "%s_%d" % (maps["idx2event"][s[0]], s[1]))
    return out


def get_n_instruments(symbols):
    """utils.py:143-148."""
    parts = [s.split("_") for s in symbols]
    return len(set(p[1] for p in parts if len(p) == 3))
