"""Battle-string parser: host-side mirror of the reference's `parse_battle(str, seed) -> Input`
(pyoak surface cpp/src/pyoak.cc:456-466, implementation cpp/include/util/parse.h:14-282).

"starmie seismictoss 101hp slp3 | snorlax bodyslam 1hp (conf:2)" -> (battle[384], durations[8])
with turn = 1 and the first listed Pokemon of each side already active (parse.h:263-282).
Pure host code (the reference's is too); state bytes follow cpp/include/libpkmn/layout.h.
"""
import numpy as np

from . import gamedata as G

_BOOST = G.DATA["boosts"]


def _put16(buf, off, v):
    buf[off] = v & 0xFF
    buf[off + 1] = (v >> 8) & 0xFF


def _parse_set(words):
    """parse.h:14-110 -> dict(species, moves[4], pp[4], hp, percent, status, sleeps, level)."""
    s = dict(species=G.match_species(words[0]), moves=[0, 0, 0, 0], pp=[64, 64, 64, 64], hp=-1,
             percent=100, status=0, sleeps=0, level=100)
    n_moves = 0
    for word in words[1:]:
        if n_moves < 4:
            mp = word.split(":")
            m = G.match_move(mp[0])
            if m is not None:
                s["moves"][n_moves] = m
                s["pp"][n_moves] = min(255, int(mp[1])) if len(mp) > 1 else 0xFF
                n_moves += 1
                continue
        lower = word.lower()
        if lower.endswith("%"):
            s["percent"] = int(lower[:-1])
        elif lower.endswith("hp"):
            s["hp"] = int(lower[:-2])
        if lower == "par":
            s["status"] = 0x40
        elif lower == "frz":
            s["status"] = 0x20
        elif lower == "psn":
            s["status"] = 0x08
        elif lower == "brn":
            s["status"] = 0x10
        elif lower.startswith("slp"):
            k = int(lower[3:])
            if k >= 7:
                raise RuntimeError("parse_set(): Invalid turns slept (must be [0, 6]): %d" % k)
            s["status"] = 7  # Status::Sleep7, hidden counter resampled per playout
            s["sleeps"] = k + 1
        elif lower.startswith("rst"):
            h = int(lower[3:])
            if h > 3 or h == 0:
                raise RuntimeError("parse_set(): Invalid sleep duration for rest (must be [1, 3]): %d" % h)
            s["status"] = 0x80 | h
        if lower.startswith("lvl"):
            s["level"] = int(lower[3:])
    return s


def _init_pokemon(s):
    """init.h:90-142 on a 24-byte slot."""
    pk = np.zeros(24, dtype=np.uint8)
    sp = s["species"]
    pk[21] = sp
    if sp == 0:
        return pk
    lvl = s["level"]
    pk[23] = lvl
    base = G.SPECIES[sp - 1]
    stats = [G.compute_stat(base[0], True, lvl)] + [G.compute_stat(base[i], False, lvl) for i in (1, 2, 3, 4)]
    for i, v in enumerate(stats):
        _put16(pk, 2 * i, v)
    for m in range(4):
        pk[10 + 2 * m] = s["moves"][m]
        pk[11 + 2 * m] = min(s["pp"][m], G.MAX_PP[s["moves"][m]]) if s["moves"][m] else 0
    hp = stats[0] * s["percent"] // 100
    if s["hp"] >= 0:
        hp = s["hp"]
    _put16(pk, 18, hp)
    pk[20] = s["status"]
    pk[22] = base[5] | (base[6] << 4)
    return pk


def _boost(stat, b):
    num, den = _BOOST[b + 6]
    return min(999, stat * num // den)


def _parse_active(pk, words):
    """parse.h:121-240: switch_in(pokemon) + boosts / explicit stats / volatiles / durations."""
    act = np.zeros(32, dtype=np.uint8)
    act[0:10] = pk[0:10]
    act[10] = pk[21]
    act[11] = pk[22]
    act[24:32] = pk[10:18]
    vol = 0
    dur = 0
    explicit = {}
    boosts = {"atk": 0, "def": 0, "spe": 0, "spc": 0}
    for word in words:
        lower = word.lower()
        for name in ("atk", "def", "spe", "spc"):
            if lower.startswith(name + "="):
                explicit[name] = int(lower[4:])
            elif lower.startswith(name):
                try:
                    boosts[name] = int(lower[3:])
                except ValueError:
                    pass
        if lower in ("(leech-seed)", "(leechseed)", "(leech)"):
            vol |= 1 << 13
        if lower in ("(invuln)", "(invulnerable)", "(dig)", "(fly)"):
            vol |= 1 << 6
        if lower in ("(lightscreen)", "(light-screen)", "(ls)"):
            vol |= 1 << 15
        if lower == "(reflect)":
            vol |= 1 << 16

        def colon(start):
            w = lower[:-1] if lower.endswith(")") else lower
            if w.startswith(start):
                parts = w.split(":")
                if len(parts) >= 2:
                    return int(parts[1])
            return -1

        c = colon("(conf")
        if c >= 0:
            if c == 0 or c > 5:
                raise RuntimeError("parse_active(): Confusion duration must be [1, 5]")
            vol |= 1 << 7
            dur = (dur & ~(7 << 18)) | (c << 18)
        for key in ("(thrash", "(petal"):
            t = colon(key)
            if t >= 0:
                vol |= 1 << 1
                dur = (dur & ~(7 << 25)) | ((t & 7) << 25)
    order = ["atk", "def", "spe", "spc"]
    bytes_ = [0, 0]
    for i, name in enumerate(order):
        off = 2 + 2 * i
        stat = int(act[off]) | (int(act[off + 1]) << 8)
        stat = _boost(stat, boosts[name])
        if name in explicit and explicit[name]:
            stat = explicit[name]
        _put16(act, off, stat)
        bytes_[i // 2] |= (boosts[name] & 15) << (4 * (i % 2))
    act[12] = bytes_[0]
    act[13] = bytes_[1]
    for k in range(8):
        act[16 + k] = (vol >> (8 * k)) & 0xFF
    return act, dur


def _parse_side(text):
    set_strings = [s for s in (x.strip() for x in text.split(";")) if s]
    if len(set_strings) == 0 or len(set_strings) > 6:
        raise RuntimeError("parse_side(): %d set given. [1, 6] required." % len(set_strings))
    sets = [_parse_set(s.split()) for s in set_strings]
    side = np.zeros(184, dtype=np.uint8)
    for i, s in enumerate(sets):
        pk = _init_pokemon(s)
        side[24 * i:24 * i + 24] = pk
        hp = int(pk[18]) | (int(pk[19]) << 8)
        if i == 0 or hp:
            side[176 + i] = i + 1
    act, dur = _parse_active(side[0:24], set_strings[0].split())
    side[144:176] = act
    for i, s in enumerate(sets):  # init_sleeps, init.h:156-165
        if (s["status"] & 7) and not (s["status"] & 0x80):
            dur = (dur & ~(7 << (3 * i))) | ((s["sleeps"] & 7) << (3 * i))
    return side, dur


def parse_battle(battle_string, seed=0x123456):
    """Returns (battle uint8[384], durations uint8[8]); raises RuntimeError like the reference."""
    sides = battle_string.split("|")
    if len(sides) != 2:
        raise RuntimeError("parse_battle(): must have two sides, delineated by '|'")
    battle = np.zeros(384, dtype=np.uint8)
    durations = np.zeros(8, dtype=np.uint8)
    for i, txt in enumerate(sides):
        side, dur = _parse_side(txt)
        battle[184 * i:184 * (i + 1)] = side
        for k in range(4):
            durations[4 * i + k] = (dur >> (8 * k)) & 0xFF
    battle[368] = 1  # turn = 1
    for k in range(8):
        battle[376 + k] = (seed >> (8 * k)) & 0xFF
    return battle, durations


def result_from_state(battle):
    """PKMN::result(battle), pkmn.h:235-272."""
    def alive(s):
        return any(int(battle[184 * s + 24 * i + 18]) | int(battle[184 * s + 24 * i + 19]) for i in range(6))

    def active_fainted(s):
        idx = int(battle[184 * s + 176]) - 1
        return (int(battle[184 * s + 24 * idx + 18]) | int(battle[184 * s + 24 * idx + 19])) == 0
    a1, a2 = alive(0), alive(1)
    if not a1:
        return 3 if not a2 else 2
    if not a2:
        return 1
    f1, f2 = active_fainted(0), active_fainted(1)
    if f1:
        return (2 << 4) | ((2 if f2 else 0) << 6)
    if f2:
        return 2 << 6
    return (1 << 4) | (1 << 6)


_VOL_FLAGS = ("bide", "thrashing", "multi-hit", "flinch", "charging", "binding", "invulnerable", "confused", "mist", "focus-energy",
              "substitute", "recharging", "rage", "leech-seed", "toxic", "light-screen", "reflect", "transform")
_STATUS_TEXT = {0x08: "PSN", 0x10: "BRN", 0x20: "FRZ", 0x40: "PAR", 0x88: "TOX"}


def _status_text(st):
    """PKMN::status_string (libpkmn/strings.h:85-112)."""
    if st == 0:
        return ""
    if st & 7:
        return "RST" if st & 0x80 else "SLP"
    return _STATUS_TEXT.get(st, "")


def battle_string(battle, durations):
    """PKMN::battle_data_to_string (libpkmn/strings.h:187-303), pyoak's `battle_string(input)` (pyoak.cc:480-485): the text
    dump of both sides -- per side the lead's changed stats, its public durations and volatiles, then one line per party
    slot in battle order."""
    b = np.frombuffer(bytes(battle), dtype=np.uint8)
    d = np.frombuffer(bytes(durations), dtype=np.uint8)
    u16 = lambda a, o: int(a[o]) | (int(a[o + 1]) << 8)
    out = []
    for s in range(2):
        side = b[184 * s:184 * (s + 1)]
        dur = int.from_bytes(bytes(d[4 * s:4 * s + 4]), "little")
        act = side[144:176]
        vol = int.from_bytes(bytes(act[16:24]), "little")
        for i in range(6):
            pid = int(side[176 + i])
            if pid == 0:
                continue
            pk = side[24 * (pid - 1):24 * pid]
            if i == 0:
                changed = False
                for k, name in ((1, "atk"), (2, "def"), (3, "spe"), (4, "spc")):
                    if u16(act, 2 * k) != u16(pk, 2 * k):
                        out.append("(%s %d>>%d) " % (name, u16(pk, 2 * k), u16(act, 2 * k)))
                        changed = True
                if changed:
                    out.append("\n")
                any_dur = False
                for text, sh, bits in (("conf: ", 18, 3), (" disable: ", 21, 4), (" attacking: ", 25, 3), (" binding: ", 28, 3)):
                    v = (dur >> sh) & ((1 << bits) - 1)
                    if v:
                        out.append("%s%d" % (text, v))
                        any_dur = True
                if any_dur:
                    out.append("\n")
                vt = "".join("(%s)" % n for k, n in enumerate(_VOL_FLAGS) if (vol >> k) & 1)
                for name, sh, mask in (("confusion_left", 18, 7), ("attacks", 21, 7), ("state", 24, 0xFFFF), ("sub_hp", 40, 0xFF)):
                    if (vol >> sh) & mask:
                        vt += "(%s: %d)" % (name, (vol >> sh) & mask)
                if (vol >> 48) & 15:
                    vt += "(transform: %s)" % G.SPECIES_NAMES[(vol >> 48) & 15]
                for name, sh, mask in (("disable_left", 52, 15), ("disable_move", 56, 7), ("toxic_counter", 59, 31)):
                    if (vol >> sh) & mask:
                        vt += "(%s: %d)" % (name, (vol >> sh) & mask)
                if vt:
                    out.append(vt + "\n")
            else:
                out.append("  ")
            out.append(G.SPECIES_NAMES[int(pk[21])])
            if int(pk[23]) != 100:
                out.append(" L%d" % int(pk[23]))
            out.append(": ")
            hp, mx = u16(pk, 18), u16(pk, 0)
            if hp == 0:
                out.append("KO \n")
                continue
            pct = int(np.ceil(np.float32(np.float32(100) * np.float32(hp)) / np.float32(mx)))     # Pokemon::percent, data.h:49-51
            out.append("%d%% (%d/%d) " % (pct, hp, mx))
            st = int(pk[20])
            if st:
                out.append(_status_text(st))
                if st & 7:
                    out.append(":%d" % ((st & 7) if st & 0x80 else (dur >> (3 * i)) & 7))
                out.append(" ")
            for m in range(4):
                out.append("%s:%d " % (G.MOVE_NAMES[int(pk[10 + 2 * m])], int(pk[11 + 2 * m])))
            out.append("\n")
        if s == 0:
            out.append("--- --- --- %d --- --- ---\n" % u16(b, 368))
    return "".join(out)


def choice_label(side, choice):
    """PKMN::side_choice_string (libpkmn/strings.h:30-51); side = 184 bytes."""
    kind, data = int(choice) & 3, int(choice) >> 2
    slot = lambda k: side[24 * (int(side[176 + k - 1]) - 1):][:24]
    if kind == 0:
        return "Pass"
    if kind == 1:
        return "None" if data == 0 else G.MOVE_NAMES[int(slot(1)[8 + 2 * data])]
    return G.SPECIES_NAMES[int(slot(data)[21])]


def format_output(battle, output):
    """MCTS::output_string (util/strings.h:61-152), pyoak's `format(input, output)` (pyoak.cc:487-492).  `output`: the dict
    oak_amd.search.tree_search returns.  (The reference's function cannot be run here -- util/strings.h reaches Eigen -- so this
    text layout is a port by reading, unlike battle_string which is checked against the reference's own output.)"""
    b = np.frombuffer(bytes(battle), dtype=np.uint8)
    m, n = int(output["m"]), int(output["n"])
    l1 = [choice_label(b[0:184], c) for c in output["p1_choices"][:m]]
    l2 = [choice_label(b[184:368], c) for c in output["p2_choices"][:n]]
    fix = lambda label: "%-8s" % label[:7]
    row = lambda prefix, cells: prefix + "".join(c + "  " for c in cells) + "\n"
    num = lambda arr, k: ["%-8.3f" % float(x) for x in arr[:k]]
    out = ["Iterations: %d, Time: %g ms\n" % (int(output["iterations"]), float(output["duration_ms"])),
           "Value: %.3f\n\n" % float(output["empirical_value"])]
    for name, labels, k, side in (("Player 1:", l1, m, "p1"), ("Player 2:", l2, n, "p2")):
        out.append(name + "\n")
        out.append(row("   ", ["%-8s" % x for x in labels]))
        out.append(row("e: ", num(output[side + "_empirical"], k)))
        out.append(row("n: ", num(output[side + "_nash"], k)))
        out.append(row("p: ", num(output[side + "_prior"], k)))
    visits, values = np.asarray(output["visit_matrix"]), np.asarray(output["value_matrix"])
    header = fix(" " * 8) + " " + "".join(fix(x) + " " for x in l2) + "\n"
    out.append("\nEV Matrix:\n" + header)
    for i in range(m):
        out.append(fix(l1[i]) + " " + "".join(" ----    " if visits[i, j] == 0 else "%-8.3f " % (values[i, j] / visits[i, j]) for j in range(n)) + "\n")
    out.append("\nVisits:\n" + header)
    for i in range(m):
        out.append(fix(l1[i]) + " " + "".join(" ----    " if visits[i, j] == 0 else "%-8d " % int(visits[i, j]) for j in range(n)) + "\n")
    return "".join(out)
