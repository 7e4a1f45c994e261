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
