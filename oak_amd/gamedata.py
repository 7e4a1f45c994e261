"""Gen-1 game data tables (moves, species, type chart, OU learnsets) and name lookups.

Data file: oak_amd/data/gen1_data.json (generated once from the reference's data headers
by tools/extract_reference_data.py; public game facts).  Mirrors what the reference keeps
in cpp/include/libpkmn/data/{moves,species,types}.h and cpp/include/format/ou/data.h:12-36.
"""
import json
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "data", "gen1_data.json")) as _f:
    DATA = json.load(_f)

MOVE_NAMES = DATA["move_names"]          # index = move id (0 = None ... 165 = Struggle)
SPECIES_NAMES = DATA["species_names"]    # index = species id (0 = None ... 151 = Mew)
TYPE_NAMES = DATA["type_names"]
EFFECT_NAMES = DATA["effect_names"]
MOVES = DATA["moves"]                    # [effect, bp, type, accuracy, target, pp] for id-1
MAX_PP = [0] + DATA["max_pp"]            # min(pp/5*8, 61), moves.h:1794-1796
SPECIES = DATA["species"]                # [hp, atk, def, spe, spc, type1, type2] for id-1

_MOVE_BY_LOWER = {n.lower(): i for i, n in enumerate(MOVE_NAMES)}
_SPECIES_BY_LOWER = {n.lower(): i for i, n in enumerate(SPECIES_NAMES)}


def move_id(name):
    return _MOVE_BY_LOWER[name.lower().replace(" ", "").replace("-", "")]


def species_id(name):
    return _SPECIES_BY_LOWER[name.lower().replace(" ", "").replace("-", "")]


def _unique_prefix(names, width, token):
    """PKMN::unique_index over the fixed-width name arrays (libpkmn/strings.h:53-83; widths data/strings.h:17-19,63-66): a
    token matches every name it is a case-insensitive PREFIX of; exactly one match -> its index, anything else -> -1.  The
    reference's "prefer the exact-length match" branch compares with the ARRAY width (13 / 12), which no token that matched
    can have, so an ambiguous prefix is never rescued by an exact name: "mew" (Mewtwo, Mew) and "thunder" (ThunderShock,
    Thunderbolt, ThunderWave, Thunder) match nothing.  A drop-in does what the code does."""
    t = token.lower()
    if len(t) >= width:
        return -1
    hits = [i for i, n in enumerate(names) if n.lower().startswith(t)]
    return hits[0] if len(hits) == 1 else -1


def match_move(token):
    """PKMN::string_to_move as parse_set uses it (util/parse.h:24-41): move id, or None when the word is not (uniquely) a move."""
    i = _unique_prefix(MOVE_NAMES, 13, token)
    return None if i < 0 else i


def match_species(token):
    """PKMN::string_to_species (libpkmn/strings.h:313-321); raises like the reference."""
    i = _unique_prefix(SPECIES_NAMES, 12, token)
    if i < 0:
        raise RuntimeError("Could not match string to Species")
    return i


def ou_pools():
    """(legal_species u8[149], pool_moves u8[152,48], pool_sizes u8[152])."""
    legal = np.array(DATA["ou_legal_species"], dtype=np.uint8)
    pools = np.zeros((152, 48), dtype=np.uint8)
    sizes = np.zeros(152, dtype=np.uint8)
    for k, v in DATA["ou_move_pools"].items():
        s = int(k)
        sizes[s] = len(v)
        pools[s, :len(v)] = v
    return legal, pools, sizes


def compute_stat(base, hp=False, level=100):
    """init.h:35-40 with DVs 15 and max stat exp."""
    core = 2 * (base + 15) + 63
    return core * level // 100 + ((level + 10) if hp else 5)
