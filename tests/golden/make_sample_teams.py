#!/usr/bin/env python3
"""Writes tests/golden/ou_sample_teams.json: the 16 Smogon sample teams the reference's programs default to
(`Teams::ou_sample_teams`, cpp/include/teams/ou-sample-teams.h) as DATA -- per team six [species, move x 4] name lists in the
header's own order and spelling.  Run here (the reference checkout is read as text); the JSON is what travels."""
import json
import os
import re

SRC = "/root/reference/cpp/include/teams/ou-sample-teams.h"
text = open(SRC).read()
sets = re.findall(r"Set\{Species::(\w+),\s*\{([^}]*)\}\}", text)
rows = [[sp] + [m.strip() for m in moves.split(",") if m.strip()] for sp, moves in sets]
assert len(rows) == 16 * 6 and all(len(r) == 5 for r in rows), (len(rows), [r for r in rows if len(r) != 5])
teams = [rows[6 * i:6 * i + 6] for i in range(16)]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ou_sample_teams.json")
json.dump({"source": "cpp/include/teams/ou-sample-teams.h (Teams::ou_sample_teams), names as spelled there", "teams": teams}, open(out, "w"), indent=0)
print("wrote", out, len(teams), "teams")
