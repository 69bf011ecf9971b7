"""Token stream -> Standard MIDI File, without pretty_midi.

Semantics of the reference's data/data_processing_reverse.py:12-53 (tuples_to_mid): a cursor advances by
TIMESHIFT tokens (milliseconds); ON_<INSTR> opens a note at the cursor (re-opening overwrites the start),
OFF_<INSTR> closes it (an OFF without a matching ON is ignored); special tokens are skipped.  Five tracks with
fixed programs and velocities: DRUMS (program 0, percussion channel), PIANO 0, GUITAR 24, BASS 32, STRINGS 48.

File layout = what pretty_midi's defaults produce: format 1, 220 ticks per quarter note at 120 bpm
(1 s = 440 ticks), track 0 carries the tempo, one track per instrument (name, program change, notes), drums on
channel 9, the others on channels 0, 1, 2, ... skipping 9.  read_midi() is the inverse used by the tests.
Pure integer/struct work on the host; nothing here touches the GPU path."""
import struct

PROGRAMS = {"DRUMS": (0, True), "PIANO": (0, False), "GUITAR": (24, False), "BASS": (32, False), "STRINGS": (48, False)}
VELOCITIES = {"BASS": 127, "DRUMS": 120, "GUITAR": 95, "PIANO": 110, "STRINGS": 85}
RESOLUTION = 220                      # ticks per quarter note
TEMPO_US = 500000                     # microseconds per quarter note (120 bpm)
TICKS_PER_SECOND = RESOLUTION * 1_000_000 / TEMPO_US


def symbols_to_notes(symbols, verbose=False):
    """symbols: strings such as 'ON_PIANO_60', 'TIMESHIFT_120', '<START>'  ->  {instrument: [(start_s, end_s, pitch)]}."""
    notes = {k: [] for k in PROGRAMS}
    active = {}
    cursor = 0.0
    for sym in symbols:
        if sym[0] == "<":
            continue
        parts = sym.split("_")
        if parts[0] == "TIMESHIFT":
            cursor += float(parts[1]) / 1000.0
            continue
        on_off, instrument, pitch = parts[0], parts[1], int(parts[2])
        if on_off == "ON":
            active[(instrument, pitch)] = cursor
        elif (instrument, pitch) in active:
            notes[instrument].append((active[(instrument, pitch)], cursor, pitch))
        elif verbose:
            print("Ignoring %s %d: no previous ON event" % (sym, pitch))
    return notes


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track(events):
    """events: list of (tick, order, bytes) -> MTrk chunk (delta times, end-of-track appended)."""
    events = sorted(events, key=lambda e: (e[0], e[1]))
    body, last = b"", 0
    for tick, _, data in events:
        body += _vlq(tick - last) + data
        last = tick
    body += _vlq(0) + b"\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(body)) + body


def notes_to_midi_bytes(notes):
    tracks = [_track([(0, 0, b"\xff\x51\x03" + struct.pack(">I", TEMPO_US)[1:])])]
    channel = 0
    for name, (program, is_drum) in PROGRAMS.items():
        if is_drum:
            ch = 9
        else:
            if channel == 9:
                channel += 1
            ch = channel
            channel += 1
        label = name.lower().encode()
        ev = [(0, 0, b"\xff\x03" + _vlq(len(label)) + label), (0, 1, bytes([0xC0 | ch, program]))]
        for start, end, pitch in notes.get(name, []):
            t0, t1 = int(round(start * TICKS_PER_SECOND)), int(round(end * TICKS_PER_SECOND))
            ev.append((t0, 3, bytes([0x90 | ch, pitch, VELOCITIES[name]])))
            ev.append((t1, 2, bytes([0x80 | ch, pitch, 0])))             # note-off sorts before a note-on at the same tick
        tracks.append(_track(ev))
    head = b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), RESOLUTION)
    return head + b"".join(tracks)


def symbols_to_midi_bytes(symbols, verbose=False):
    return notes_to_midi_bytes(symbols_to_notes(symbols, verbose=verbose))


def write_midi(path, symbols, verbose=False):
    with open(path, "wb") as fh:
        fh.write(symbols_to_midi_bytes(symbols, verbose=verbose))


def read_midi(data):
    """Minimal SMF reader (what notes_to_midi_bytes writes): -> (resolution, tempo_us, {track_name: {'program': p,
    'channel': c, 'notes': [(start_tick, end_tick, pitch, velocity)]}})."""
    assert data[:4] == b"MThd"
    _, fmt, ntrk, res = struct.unpack(">IHHH", data[4:14])
    pos, tempo, out = 14, None, {}
    for _ in range(ntrk):
        assert data[pos:pos + 4] == b"MTrk"
        n = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + n]
        pos += 8 + n
        i, tick, name, info, open_notes = 0, 0, None, {"program": None, "channel": None, "notes": []}, {}
        while i < len(body):
            delta = 0
            while True:
                b = body[i]
                i += 1
                delta = (delta << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            tick += delta
            st = body[i]
            if st == 0xFF:
                kind, ln = body[i + 1], body[i + 2]
                payload = body[i + 3:i + 3 + ln]
                i += 3 + ln
                if kind == 0x51:
                    tempo = int.from_bytes(payload, "big")
                elif kind == 0x03:
                    name = payload.decode()
            elif st & 0xF0 == 0xC0:
                info["program"], info["channel"] = body[i + 1], st & 0x0F
                i += 2
            elif st & 0xF0 == 0x90:
                open_notes[body[i + 1]] = (tick, body[i + 2])
                i += 3
            elif st & 0xF0 == 0x80:
                if body[i + 1] in open_notes:
                    t0, vel = open_notes.pop(body[i + 1])
                    info["notes"].append((t0, tick, body[i + 1], vel))
                i += 3
            else:
                raise ValueError("unexpected status byte %02x" % st)
        if name is not None:
            out[name] = info
    return res, tempo, out
