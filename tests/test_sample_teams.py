"""tests/golden/ou_sample_teams.json (the 16 sample teams the reference's programs default to, written from its header by
tests/golden/make_sample_teams.py): every name resolves through the product's own name matching (gamedata.match_species / match_move:
unique case-insensitive prefixes like the reference's PKMN::string_to_species / string_to_move), teams are complete, and -- when the
reference checkout is here -- the file still equals what the script extracts from teams/ou-sample-teams.h."""
import json
import os
import re

import pytest

from oak_amd import gamedata as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "ou_sample_teams.json")


def test_sample_teams_resolve_and_are_complete():
    teams = json.load(open(FIX))["teams"]
    assert len(teams) == 16
    seen = set()
    for t in teams:
        assert len(t) == 6
        species = [G.match_species(s[0]) for s in t]
        assert all(species) and len(set(species)) == 6, t
        for s in t:
            moves = [G.match_move(m) for m in s[1:]]
            assert len(s) == 5 and all(moves) and len(set(moves)) == 4, s
        seen.add(json.dumps(t))
    assert len(seen) == 16, "two sample teams are identical"
    assert G.match_species("Tauros") and all("Tauros" in [s[0] for s in t] for t in teams[:6])   # (every classic team carries one)


def test_sample_teams_fixture_matches_the_reference_header():
    src = "/root/reference/cpp/include/teams/ou-sample-teams.h"
    if not os.path.exists(src):
        pytest.skip("reference checkout absent (the fixture is what travels)")
    sets = re.findall(r"Set\{Species::(\w+),\s*\{([^}]*)\}\}", open(src).read())
    rows = [[sp] + [m.strip() for m in moves.split(",") if m.strip()] for sp, moves in sets]
    assert [rows[6 * i:6 * i + 6] for i in range(16)] == json.load(open(FIX))["teams"]
