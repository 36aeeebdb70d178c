"""Map tables and per-map rule constants of the SSD grid worlds.

Data restated from the reference (src/envs/ssd/constants.py:13-116 for the ASCII layouts; src/envs/ssd/cleanup.py:31-54
for the Cleanup thresholds; src/envs/ssd/harvest.py:20-22 for the Harvest regrowth table).  The alphabet is the
reference's: '@' wall, 'P' spawn point, 'B' Cleanup apple site, 'A' Harvest apple site, 'H' waste, 'R' river,
'S' stream, ' ' empty.  The native library derives every site list from the layout by a row-major scan, exactly as
the reference does (cleanup.py:77-90, harvest.py:31-35), so any layout over this alphabet is accepted.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

# One 12-row block of the Cleanup "default5" map; default5 = border + 2 blocks (23 rows) + border,
# default10 = border + rows of default5 twice (46 rows) + border.
_CLEANUP_ROWS_A = (
    "@RRRRRR     BBBBB@",
    "@HHHHHH    P BBBB@",
    "@RRRRRR     BBBBB@",
    "@RRRRR       BBBB@",
    "@RRRRR      BBBBB@",
    "@HHHH P      BBBB@",
    "@RRRRR      BBBBB@",
    "@HHHHHHSSSSSSBBBB@",
    "@HHHHHHSSSSSSBBBB@",
    "@RRRRR       BBBB@",
    "@HHHHH      BBBBB@",
    "@RRRRRR    P BBBB@",
    "@HHHHHH     BBBBB@",
    "@RRRRR       BBBB@",
    "@HHHH       BBBBB@",
    "@RRRRR     P BBBB@",
    "@HHHHH      BBBBB@",
    "@RRRRR       BBBB@",
    "@HHHH P     BBBBB@",
    "@RRRRR       BBBB@",
    "@HHHHH      BBBBB@",
    "@RRRRR       BBBB@",
    "@HHHH       BBBBB@",
)
_BORDER18 = "@" * 18

CLEANUP_DEFAULT3 = (
    "@@@@@@@@@@",
    "@HH   P B@",
    "@RR    BB@",
    "@HH     B@",
    "@RR    BB@",
    "@HH P   B@",
    "@RR    BB@",
    "@HH     B@",
    "@RRP   BB@",
    "@@@@@@@@@@",
)
CLEANUP_DEFAULT5 = (_BORDER18,) + _CLEANUP_ROWS_A + (_BORDER18,)
CLEANUP_DEFAULT10 = (_BORDER18,) + _CLEANUP_ROWS_A + _CLEANUP_ROWS_A + (_BORDER18,)

HARVEST_DEFAULT10 = (
    "@" * 38,
    "@ P   P           P          P    P  @",
    "@        A   AA         AAA    A     @",
    "@     A AAA  AAA    A    A AA AAAA   @",
    "@    AAA A    A  A AAA  A  A   A A   @",
    "@    A A       AAA A  AAA            @",
    "@      AAA  AAA  A      AAA   AAA    @",
    "@   P      P          P      P   P   @",
    "@" * 38,
)


@dataclass(frozen=True)
class MapSpec:
    env: str                      # "cleanup" | "harvest"
    rows: Tuple[str, ...]
    # Cleanup rule constants (cleanup.py:31-54)
    threshold_depletion: float = 0.0
    threshold_restoration: float = 0.0
    waste_spawn_prob: float = 0.0
    apple_respawn_prob: float = 0.0
    # Harvest regrowth table indexed by min(#neighbour apples, 3) (harvest.py:20-22,118)
    harvest_spawn_prob: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 0.0)

    @property
    def height(self) -> int:
        return len(self.rows)

    @property
    def width(self) -> int:
        return len(self.rows[0])

    @property
    def ascii(self) -> bytes:
        assert all(len(r) == self.width for r in self.rows), "ragged map"
        return "".join(self.rows).encode("ascii")


def cleanup_spec(map_name: str) -> MapSpec:
    """CleanupEnv.__init__ map switch (cleanup.py:31-54): unknown names fall back to the N5 layout."""
    if map_name == "default3":
        return MapSpec("cleanup", CLEANUP_DEFAULT3, 0.4, 0.0, 0.5, 0.3)
    if map_name == "default10":
        return MapSpec("cleanup", CLEANUP_DEFAULT10, 0.99, 0.0, 0.5, 0.05)
    return MapSpec("cleanup", CLEANUP_DEFAULT5, 0.99, 0.0, 0.5, 0.05)


def harvest_spec(map_name: str) -> MapSpec:
    """HarvestEnv.__init__ (harvest.py:18-22): SPAWN_PROB only exists for map == "default10"; any other name makes
    the reference raise AttributeError at the first regrowth (harvest.py:118), so it is rejected here."""
    if map_name != "default10":
        raise AttributeError("'HarvestEnv' object has no attribute 'SPAWN_PROB' (map must be 'default10', "
                             "reference harvest.py:20-22,118)")
    return MapSpec("harvest", HARVEST_DEFAULT10, harvest_spawn_prob=(0.0, 0.05, 0.08, 0.1))


def get_spec(env: str, map_name: str) -> MapSpec:
    if env == "cleanup":
        return cleanup_spec(map_name)
    if env == "harvest":
        return harvest_spec(map_name)
    raise KeyError(env)


# Colour tables for the observation LUT, restated from map_env.py:33-62 (DEFAULT_COLOURS) and cleanup.py:14-17.
AGENT_COLOURS: Dict[int, Tuple[int, int, int]] = {
    1: (159, 67, 255), 2: (2, 81, 154), 3: (204, 0, 204), 4: (216, 30, 54), 5: (254, 151, 0),
    6: (205, 155, 155), 7: (99, 99, 255), 8: (250, 204, 255), 9: (238, 223, 16),
}
