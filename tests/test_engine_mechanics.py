"""CPU: gen-1 mechanics of the oracle engine on hand-built positions.  The libpkmn source is absent from the
reference checkout, so these pin the restatement to well-documented gen-1 / Pokemon
Showdown behaviours -- the same facts cpp/src/search-test.cc relies on for sleep and confusion -- and freeze
the decisions documented at the top of oracle/gen1_engine.c.  Positions use the reference's battle-string
syntax (util/parse.h); damage rolls are pinned through the calc overrides (mcts.h:575-588) where needed."""
import numpy as np

import oracle_lib as O
from oak_amd import gamedata as G
from oak_amd.parse import parse_battle, result_from_state

MAXROLL = np.zeros(16, np.uint8)
MAXROLL[0] = MAXROLL[8] = 255


def mv(b, side, name):
    """choice byte selecting move `name` of the active Pokemon of `side`."""
    mid = G.move_id(name)
    for k in range(4):
        if b[184 * side + 144 + 24 + 2 * k] == mid:
            return ((k + 1) << 2) | 1
    raise KeyError(name)


def hp(b, side, idx=None):
    if idx is None:
        idx = int(b[184 * side + 176]) - 1
    o = 184 * side + 24 * idx + 18
    return int(b[o]) | (int(b[o + 1]) << 8)


def status(b, side):
    idx = int(b[184 * side + 176]) - 1
    return int(b[184 * side + 24 * idx + 20])


def vol(b, side):
    return int.from_bytes(bytes(b[184 * side + 160:184 * side + 168]), "little")


def astat(b, side, k):  # 0 hp 1 atk 2 def 3 spe 4 spc
    o = 184 * side + 144 + 2 * k
    return int(b[o]) | (int(b[o + 1]) << 8)


def step(b, d, c1, c2, seed=None, overrides=MAXROLL):
    opt = O.Options(d)
    opt.set(None, overrides)
    if seed is not None:
        b[376:384] = np.array([seed], dtype=np.uint64).view(np.uint8)
    r = O.update(b, c1, c2, opt)
    d[:] = opt.durations
    return r


def find_seed(pos, c1, c2, pred, tries=400):
    """first battle seed for which pred(battle_after, result) holds (rolls are seed-determined)."""
    for s in range(tries):
        b, d = parse_battle(pos, 1000 + s)
        r = step(b, d, c1(b), c2(b))
        if pred(b, r):
            return b, d, r
    raise AssertionError("no seed satisfies predicate")


def test_faster_pokemon_moves_first_and_ko_ends_the_turn():
    b, d = parse_battle("starmie surf | rhydon earthquake 1hp")
    r = step(b, d, mv(b, 0, "Surf"), mv(b, 1, "Earthquake"))
    assert hp(b, 1) == 0 and hp(b, 0) == hp(parse_battle("starmie surf | rhydon earthquake")[0], 0)  # Rhydon never moved
    assert r == 1  # WIN: last Pokemon fainted


def test_quick_attack_priority_and_counter_moves_last():
    b, d = parse_battle("snorlax quickattack | jolteon thunderbolt 1hp")
    assert step(b, d, mv(b, 0, "QuickAttack"), mv(b, 1, "Thunderbolt")) == 1
    b, d = parse_battle("alakazam counter | snorlax bodyslam")
    full = hp(b, 1)
    step(b, d, mv(b, 0, "Counter"), mv(b, 1, "BodySlam"))
    lost = hp(parse_battle("alakazam counter | snorlax bodyslam")[0], 0) - hp(b, 0)
    assert lost > 0 and (full - hp(b, 1) == min(full, 2 * lost))


def test_hyper_beam_recharges_unless_it_kos():
    b, d, r = find_seed("tauros hyperbeam | snorlax amnesia", lambda b: mv(b, 0, "HyperBeam"), lambda b: mv(b, 1, "Amnesia"),
                        lambda b, r: hp(b, 1) < 523)
    assert vol(b, 0) & (1 << 11)                                    # Recharging
    assert list(O.choices(b, 0, 1)) == [1]                          # forced "move 0"
    h = hp(b, 1)
    step(b, d, 1, mv(b, 1, "Amnesia"))
    assert not (vol(b, 0) & (1 << 11)) and hp(b, 1) == h            # the recharge turn does nothing
    b, d, r = find_seed("tauros hyperbeam | chansey softboiled 1hp; snorlax rest", lambda b: mv(b, 0, "HyperBeam"),
                        lambda b: mv(b, 1, "SoftBoiled"), lambda b, r: hp(b, 1) == 0)
    assert not (vol(b, 0) & (1 << 11)) and r == (2 << 6)            # KO: no recharge, P2 must switch


def test_substitute_cost_blocking_and_break():
    b, d = parse_battle("alakazam substitute | gengar hypnosis toxic")
    step(b, d, mv(b, 0, "Substitute"), mv(b, 1, "Toxic"))
    assert hp(b, 0) == 313 - 313 // 4 and (vol(b, 0) & (1 << 10)) and ((vol(b, 0) >> 40) & 0xFF) == 313 // 4 + 1
    assert status(b, 0) == 0                                        # Toxic blocked by the substitute
    b, d = parse_battle("alakazam substitute 70hp | gengar hypnosis")
    step(b, d, mv(b, 0, "Substitute"), mv(b, 1, "Hypnosis"))
    assert not (vol(b, 0) & (1 << 10)) and hp(b, 0) == 70          # not enough HP: fails


def test_explosion_faints_user_and_halves_defense():
    b, d = parse_battle("snorlax selfdestruct; tauros bodyslam | rhydon earthquake")
    r = step(b, d, mv(b, 0, "SelfDestruct"), mv(b, 1, "Earthquake"))
    assert hp(b, 0, 0) == 0 and r & 0x20                            # P1 must switch
    dmg_boom = 413 - hp(b, 1)
    # same attacker, 120 bp Mega Kick at the same (max) roll: 130 bp against HALVED defense must deal far more than 13/12 of it
    kick = None
    for s in range(200):
        b2, d2 = parse_battle("snorlax megakick | rhydon earthquake", 5000 + s)
        step(b2, d2, mv(b2, 0, "MegaKick"), mv(b2, 1, "Earthquake"))
        if 0 < 413 - hp(b2, 1) < dmg_boom:
            kick = 413 - hp(b2, 1)
            break
    assert kick is not None and dmg_boom > 1.6 * kick


def test_sleep_rest_and_wake_turn():
    b, d = parse_battle("starmie rest 100hp | snorlax seismictoss")
    step(b, d, mv(b, 0, "Rest"), mv(b, 1, "SeismicToss"))
    assert status(b, 0) == 0x82 and hp(b, 0) == 323 - 100           # rested to full, then hit
    for expect in (0x81, 0x00):
        step(b, d, mv(b, 0, "Rest"), mv(b, 1, "SeismicToss"))
        assert status(b, 0) == expect                               # two turns asleep, no move on the wake turn
    assert hp(b, 0) == 323 - 300


def test_paralysis_quarters_speed_burn_halves_attack_and_switch_restores():
    b, d = parse_battle("jolteon thunderwave; tauros bodyslam | starmie surf; snorlax rest")
    spe0 = astat(b, 1, 3)
    b, d, r = find_seed("jolteon thunderwave; tauros bodyslam | starmie surf; snorlax rest", lambda b: mv(b, 0, "ThunderWave"),
                        lambda b: mv(b, 1, "Surf"), lambda b, r: status(b, 1) == 0x40)
    assert astat(b, 1, 3) == max(spe0 // 4, 1)
    step(b, d, mv(b, 0, "ThunderWave"), (2 << 2) | 2)               # Starmie switches out
    step(b, d, mv(b, 0, "ThunderWave"), (2 << 2) | 2)               # ... and back in: penalty re-applied from status
    assert astat(b, 1, 3) == max(spe0 // 4, 1) and status(b, 1) == 0x40


def test_toxic_counter_leech_seed_and_switch_resets_to_poison():
    b, d, r = find_seed("venusaur toxic leechseed; tauros bodyslam | snorlax amnesia; chansey softboiled",
                        lambda b: mv(b, 0, "Toxic"), lambda b: mv(b, 1, "Amnesia"), lambda b, r: status(b, 1) == 0x88)
    h0 = hp(b, 1)
    assert h0 == 523 - 523 // 16                                    # first toxic tick 1/16
    step(b, d, mv(b, 0, "Toxic"), mv(b, 1, "Amnesia"), seed=5)      # Toxic fails now; tick 2/16
    assert hp(b, 1) == h0 - 2 * (523 // 16)
    step(b, d, mv(b, 0, "Toxic"), (2 << 2) | 2)                     # switch out: toxic -> regular poison
    assert int(b[184 + 20]) == 0x08


def test_binding_traps_and_counts_down():
    b, d, r = find_seed("cloyster clamp | snorlax bodyslam", lambda b: mv(b, 0, "Clamp"), lambda b: mv(b, 1, "BodySlam"),
                        lambda b, r: bool(vol(b, 0) & (1 << 5)))
    h0, hc = hp(b, 1), hp(b, 0)
    assert hc == 303                                                # Snorlax (slower) could not move
    assert (int.from_bytes(bytes(d[0:4]), "little") >> 28) & 7 == 1  # public binding counter started
    assert len(O.choices(b, 0, 1)) == 1                             # the binder is locked in
    attacks = (vol(b, 0) >> 21) & 7
    for k in range(attacks):
        step(b, d, int(O.choices(b, 0, 1)[0]), mv(b, 1, "BodySlam"))
        assert hp(b, 0) == 303
    assert not (vol(b, 0) & (1 << 5)) and hp(b, 1) < h0


def test_struggle_when_out_of_pp_and_recoil():
    b, d = parse_battle("snorlax bodyslam:0 | chansey softboiled")
    assert list(O.choices(b, 0, 1)) == [1]
    h = hp(b, 0)
    step(b, d, 1, mv(b, 1, "SoftBoiled"))
    dealt = 703 - hp(b, 1)
    assert dealt > 0 and hp(b, 0) == h - max(dealt // 2, 1)


def test_switching_clears_boosts_and_volatiles():
    b, d = parse_battle("snorlax amnesia reflect; tauros bodyslam | chansey softboiled")
    step(b, d, mv(b, 0, "Amnesia"), mv(b, 1, "SoftBoiled"))
    step(b, d, mv(b, 0, "Reflect"), mv(b, 1, "SoftBoiled"))
    assert b[144 + 13] >> 4 == 2 and (vol(b, 0) & (1 << 16))
    step(b, d, (2 << 2) | 2, mv(b, 1, "SoftBoiled"))
    assert b[144 + 12] == 0 and b[144 + 13] == 0 and vol(b, 0) == 0 and b[176] == 2


def test_reflect_halves_physical_damage_but_not_crits():
    seen = set()
    for seed in range(40):
        dm = []
        crit = None
        for pos in ("snorlax amnesia (reflect) | tauros bodyslam", "snorlax amnesia | tauros bodyslam"):
            b, d = parse_battle(pos, 3000 + seed)
            opt = O.Options(d)
            opt.set(None, MAXROLL)
            O.update(b, mv(b, 0, "Amnesia"), mv(b, 1, "BodySlam"), opt)
            dm.append(523 - hp(b, 0))
            crit = (int(opt.actions[9]) >> 2) & 3 == 2            # P2's critical_hit field (bit 10 of its 8 bytes)
        if crit:
            assert dm[0] == dm[1]                                   # crits ignore Reflect
        else:
            assert abs(dm[0] * 2 - dm[1]) <= 3 and dm[0] < dm[1]    # defense doubled
        seen.add(crit)
    assert seen == {True, False}


def test_freeze_is_permanent_until_fire():
    b, d = parse_battle("snorlax bodyslam frz | charizard ember tackle")
    for _ in range(5):
        step(b, d, mv(b, 0, "BodySlam"), mv(b, 1, "Tackle"))
        assert status(b, 0) == 0x20 and hp(b, 1) == 359            # frozen: never moves
    step(b, d, mv(b, 0, "BodySlam"), mv(b, 1, "Ember"))
    assert status(b, 0) in (0, 0x10)                                # thawed by a fire move (and may now burn later)


def test_turn_limit_is_a_tie_and_double_ko_is_a_tie():
    b, d = parse_battle("chansey softboiled | chansey softboiled")
    b[368:370] = np.array([999], dtype=np.uint16).view(np.uint8)
    assert step(b, d, mv(b, 0, "SoftBoiled"), mv(b, 1, "SoftBoiled")) == 3
    b, d = parse_battle("snorlax selfdestruct | chansey softboiled 1hp")
    assert step(b, d, mv(b, 0, "SelfDestruct"), mv(b, 1, "SoftBoiled")) == 3


def test_fainted_side_must_switch_and_turn_counter_rules():
    b, d = parse_battle("starmie surf | rhydon earthquake 1hp; chansey softboiled")
    r = step(b, d, mv(b, 0, "Surf"), mv(b, 1, "Earthquake"))
    assert r == (2 << 6) and int(b[368]) == 1                       # turn does not advance until the replacement
    assert list(O.choices(b, 1, 2)) == [(2 << 2) | 2] and list(O.choices(b, 0, 0)) == [0]
    r = step(b, d, 0, (2 << 2) | 2)
    assert r == 0x50 and int(b[368]) == 2 and int(b[184 + 176]) == 2


def test_durations_follow_party_slots_on_switch():
    b, d, r = find_seed("gengar hypnosis | snorlax rest; chansey softboiled", lambda b: mv(b, 0, "Hypnosis"),
                        lambda b: mv(b, 1, "Rest"), lambda b, r: 0 < status(b, 1) < 8)
    assert int.from_bytes(bytes(d[4:8]), "little") & 7 == 2         # put to sleep, then slept through its own move: 2
    step(b, d, mv(b, 0, "Hypnosis"), (2 << 2) | 2)
    dw = int.from_bytes(bytes(d[4:8]), "little")
    assert dw & 7 == 0 and (dw >> 3) & 7 == 2                       # counter moved with the sleeper to slot 2


# ---- second batch: formula-level and effect-level facts -------------------------------------------------------------
def actions_of(pos, c1, c2, seed):
    b, d = parse_battle(pos, seed)
    opt = O.Options(d)
    opt.set(None, MAXROLL)
    r = O.update(b, c1(b), c2(b), opt)
    return b, r, opt.actions.copy()


def crit_flag(actions, side):           # critical_hit field: bit 10 of the side's 8 action bytes (layout.h:98-117)
    return (int(actions[8 * side + 1]) >> 2) & 3 == 2


def test_damage_formula_exact_number():
    """L100 Tauros Body Slam into Starmie, max roll, no crit: floor(floor(floor(2*100/5+2) * 85 * atk / def) / 50) + 2,
    x1.5 STAB, neutral, x255/255 -- the cartridge formula, worked out here from the two stats (both divided by 4
    first because one of them exceeds 255)."""
    b0, _ = parse_battle("tauros bodyslam | starmie recover")
    atk, dfn = astat(b0, 0, 1), astat(b0, 1, 2)
    assert atk > 255 or dfn > 255
    atk, dfn = atk // 4, dfn // 4
    base = (2 * 100 // 5 + 2) * 85 * atk // dfn // 50 + 2
    expected = base + base // 2
    seen = False
    for s in range(60):
        b, r, act = actions_of("tauros bodyslam | starmie recover", lambda b: mv(b, 0, "BodySlam"), lambda b: mv(b, 1, "Recover"), 7000 + s)
        if not crit_flag(act, 0) and hp(b0, 1) - hp(b, 1) > 0:
            # Starmie (faster) recovered nothing at full HP, then took the hit
            assert hp(b0, 1) - hp(b, 1) == expected
            seen = True
    assert seen


def test_critical_hits_double_level_and_ignore_stat_stages():
    """Crit damage uses level x 2 and the UNMODIFIED stats: a -2 Attack Tauros crits for exactly what an unboosted one does."""
    b0, _ = parse_battle("tauros bodyslam | starmie recover")
    atk, dfn = astat(b0, 0, 1) // 4, astat(b0, 1, 2) // 4
    base = (2 * 200 // 5 + 2) * 85 * atk // dfn // 50 + 2
    expected = base + base // 2
    hits = 0
    for s in range(300):
        b, r, act = actions_of("tauros bodyslam (atk-2) | starmie recover", lambda b: mv(b, 0, "BodySlam"), lambda b: mv(b, 1, "Recover"), 9000 + s)
        if crit_flag(act, 0):
            assert hp(b0, 1) - hp(b, 1) == expected
            hits += 1
    # crit chance = floor(base speed / 2) / 256 = 55 / 256 for Tauros (base 110)
    assert 0.12 < hits / 300 < 0.32


def test_high_critical_moves_and_rates():
    n = 400
    slash = sum(crit_flag(actions_of("persian slash | chansey softboiled", lambda b: mv(b, 0, "Slash"), lambda b: mv(b, 1, "SoftBoiled"), 100 + s)[2], 0)
                for s in range(n))
    assert slash / n > 0.97                 # min(8 * floor(115 / 2), 255) / 256
    slow = sum(crit_flag(actions_of("snorlax bodyslam | chansey softboiled", lambda b: mv(b, 0, "BodySlam"), lambda b: mv(b, 1, "SoftBoiled"), 100 + s)[2], 0)
               for s in range(n))
    assert slow / n < 0.12                  # floor(30 / 2) / 256 = 5.9%


def test_type_immunities():
    b, d = parse_battle("snorlax bodyslam | gengar nightshade")
    step(b, d, mv(b, 0, "BodySlam"), mv(b, 1, "NightShade"))
    assert hp(b, 1) == hp(parse_battle("snorlax bodyslam | gengar nightshade")[0], 1)      # Normal vs Ghost
    b, d = parse_battle("rhydon earthquake | zapdos thunderwave")
    step(b, d, mv(b, 0, "Earthquake"), mv(b, 1, "ThunderWave"))
    assert hp(b, 1) == hp(parse_battle("rhydon earthquake | zapdos thunderwave")[0], 1)    # Ground vs Flying
    assert status(b, 0) == 0                                                                # Thunder Wave vs Ground
    b, d = parse_battle("venusaur leechseed | exeggutor psychic")
    step(b, d, mv(b, 0, "LeechSeed"), mv(b, 1, "Psychic"))
    assert not (vol(b, 1) & (1 << 13))                                                      # Leech Seed vs Grass


def test_multi_hit_distribution_and_fixed_two_hitters():
    counts = {}
    for s in range(800):
        b, r, act = actions_of("cloyster spikecannon | chansey softboiled", lambda b: mv(b, 0, "SpikeCannon"), lambda b: mv(b, 1, "SoftBoiled"), s)
        k = (int(act[5]) >> 4) & 15          # multi_hit field: bits 44-47
        if k:
            counts[k] = counts.get(k, 0) + 1
    tot = sum(counts.values())
    assert set(counts) == {2, 3, 4, 5}
    assert abs(counts[2] / tot - 3 / 8) < 0.06 and abs(counts[3] / tot - 3 / 8) < 0.06
    assert abs(counts[4] / tot - 1 / 8) < 0.05 and abs(counts[5] / tot - 1 / 8) < 0.05
    # Double Kick: exactly two equal hits
    b0, _ = parse_battle("hitmonlee doublekick | snorlax amnesia")
    for s in range(40):
        b, r, act = actions_of("hitmonlee doublekick | snorlax amnesia", lambda b: mv(b, 0, "DoubleKick"), lambda b: mv(b, 1, "Amnesia"), 300 + s)
        lost = hp(b0, 1) - hp(b, 1)
        assert lost % 2 == 0


def test_counter_only_answers_normal_and_fighting():
    b, d = parse_battle("chansey counter | starmie surf")
    h1 = hp(b, 1)
    step(b, d, mv(b, 0, "Counter"), mv(b, 1, "Surf"))
    assert hp(b, 1) == h1                                           # Water damage cannot be countered
    b, d = parse_battle("snorlax counter | chansey seismictoss")
    h0, h1 = hp(b, 0), hp(b, 1)
    step(b, d, mv(b, 0, "Counter"), mv(b, 1, "SeismicToss"))
    assert h0 - hp(b, 0) == 100 and h1 - hp(b, 1) == 200            # Fighting-type damage comes back doubled


def test_mirror_move_and_metronome():
    b, d = parse_battle("fearow mirrormove | starmie surf")
    h1 = hp(b, 1)
    step(b, d, mv(b, 0, "MirrorMove"), mv(b, 1, "Surf"))           # Starmie is faster: its Surf is mirrored
    assert hp(b, 1) < h1 and int(b[183]) == G.move_id("Surf")       # last_used_move = the copied move
    b, d = parse_battle("fearow mirrormove | snorlax bodyslam")
    h1 = hp(b, 1)
    step(b, d, mv(b, 0, "MirrorMove"), mv(b, 1, "BodySlam"))       # Pidgeot is faster: nothing to mirror yet
    assert hp(b, 1) == h1
    picked = set()
    for s in range(200):
        b, r, act = actions_of("clefable metronome | chansey softboiled", lambda b: mv(b, 0, "Metronome"), lambda b: mv(b, 1, "SoftBoiled"), s)
        picked.add(int(act[7]))                                      # metronome field: bits 56-63
    assert len(picked) > 80 and G.move_id("Metronome") not in picked and 0 not in picked and 165 not in picked


def test_disable_transform_and_mimic():
    b, d, r = find_seed("alakazam disable | snorlax bodyslam amnesia", lambda b: mv(b, 0, "Disable"), lambda b: mv(b, 1, "Amnesia"),
                        lambda b, r: (vol(b, 1) >> 56) & 7 != 0)
    slot = (vol(b, 1) >> 56) & 7
    legal = [int(c) for c in O.choices(b, 1, 1)]
    assert ((slot << 2) | 1) not in legal and len(legal) == 1        # the disabled move cannot be selected
    b, d = parse_battle("ditto transform | starmie surf recover")
    step(b, d, mv(b, 0, "Transform"), mv(b, 1, "Recover"))
    assert vol(b, 0) & (1 << 17)
    assert [astat(b, 0, k) for k in range(1, 5)] == [astat(b, 1, k) for k in range(1, 5)]      # stats copied (not HP)
    assert list(b[144 + 24:144 + 32:2]) == list(b[184 + 144 + 24:184 + 144 + 32:2])            # move ids copied
    assert [int(x) for x in b[144 + 25:144 + 33:2]] == [5 if m else 0 for m in b[144 + 24:144 + 32:2]]   # 5 PP each
    b, d, r = find_seed("clefable mimic | starmie surf", lambda b: mv(b, 0, "Mimic"), lambda b: mv(b, 1, "Surf"),
                        lambda b, r: int(b[144 + 24]) == G.move_id("Surf"))
    assert int(b[10]) == G.move_id("Mimic")                          # only the ACTIVE copy changes, the party slot keeps Mimic


def test_haze_thrash_and_jump_kick_crash():
    b, d = parse_battle("vaporeon haze (atk+2) brn | snorlax amnesia (spc+4) slp3")
    step(b, d, mv(b, 0, "Haze"), mv(b, 1, "Amnesia"))
    # both sides' stages reset; Snorlax, cured of sleep by the (faster) Haze, then moves: Amnesia from 0 to +2, not +6
    assert b[144 + 12] == 0 and b[184 + 144 + 13] == (2 << 4)
    assert status(b, 1) == 0 and status(b, 0) == 0x10                # the FOE's status is cured, the user's stays
    b, d, r = find_seed("tauros thrash | chansey softboiled", lambda b: mv(b, 0, "Thrash"), lambda b: mv(b, 1, "SoftBoiled"),
                        lambda b, r: bool(vol(b, 0) & 2))
    turns = 1
    while vol(b, 0) & 2:
        assert len(O.choices(b, 0, 1)) == 1                          # locked in
        step(b, d, int(O.choices(b, 0, 1)[0]), mv(b, 1, "SoftBoiled"))
        turns += 1
    assert turns in (3, 4) and (vol(b, 0) & (1 << 7))                # 3-4 turns of thrashing, then confusion
    b, d = parse_battle("hitmonlee highjumpkick | gengar nightshade")
    h0 = hp(b, 0)
    step(b, d, mv(b, 0, "HighJumpKick"), mv(b, 1, "NightShade"))
    assert hp(b, 0) == h0 - 100                                      # immune target: no crash damage (only Night Shade's 100)


def test_ohko_speed_rule_and_swift_accuracy():
    b, d = parse_battle("rhydon horndrill | starmie recover")
    step(b, d, mv(b, 0, "HornDrill"), mv(b, 1, "Recover"))
    assert hp(b, 1) == hp(parse_battle("rhydon horndrill | starmie recover")[0], 1)         # slower user: always fails
    miss = 0
    for s in range(200):
        b, d = parse_battle("starmie swift | chansey softboiled (eva+6)", 50 + s)
        h = hp(b, 1)
        step(b, d, mv(b, 0, "Swift"), mv(b, 1, "SoftBoiled"))
        miss += hp(b, 1) == h and h < 703
    assert miss == 0                                                  # Swift ignores accuracy / evasion
