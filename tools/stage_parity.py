"""Each operator of one encoder block on the device against the rounding-emulating oracle, stage by stage.

    python tools/stage_parity.py [GTINY|G300|G7B] [--fp8]

Prints, per operator, the relative L2 distance and the share of outputs that differ when the operator is
fed the oracle's own input (`tests/stage_tools.py`).
"""

from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "omnilingual-asr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from omnilingual_asr.weights import GEOMETRIES
    from tests.stage_tools import block_stages

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "GTINY"
    fp8 = "--fp8" in sys.argv
    print("%s, one block, %s" % (name, "fp8 projections" if fp8 else "bf16"))
    for stage, rel, frac in block_stages(GEOMETRIES[name], fp8):
        print("%-44s rel-L2 %.3e   outputs that differ %.4f%%" % (stage, rel, 100.0 * frac))


if __name__ == "__main__":
    main()
