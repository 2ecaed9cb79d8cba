"""BPE-PCFG simulator: the reference's ``BpePcfgSim`` (bpe_simulator.py:104-157) behind the same
constructor; sampling (one structure draw + one terminal draw per segment) and scoring (minimum
over the candidate structures of a password's LUDS label) run as HIP kernels over a two-level
device table (montecarlo_amd.tables.BpeFlat).

The reference's ``calc_ml2p`` returns a bare float, which its own ``parse_file`` cannot unpack
(SURVEY.md §3.3); here it returns ``(ml2p, [pwd])`` so the plugin surface works, and the float is
pinned against the reference's value."""
from typing import Dict, List, Set, Tuple

import numpy as np

from .bpeX.modelreader import read_bpe
from .lib4mc.MonteCarloParent import MonteCarlo
from .lib4mc.ProbLib import expand_1d, expand_2d
from .tables import BpeFlat

UNPARSEABLE = float(1022)
_CLASS = {"L": 0, "U": 1, "D": 2, "S": 3}


def char_class(ch: str) -> int:
    """luds' tag of one character (bpe_simulator.py:23-31) as 0..3 = L U D S."""
    if ch.isalpha():
        return 1 if ch.isupper() else 0
    return 2 if ch.isdigit() else 3


def luds(pwd: str) -> Tuple[Tuple[str, int], ...]:
    """bpe_simulator.py:17-40: runs of equal character class."""
    runs: List[Tuple[str, int]] = []
    for ch in pwd:
        tag = "LUDS"[char_class(ch)]
        if runs and runs[-1][0] == tag:
            runs[-1] = (tag, runs[-1][1] + 1)
        else:
            runs.append((tag, 1))
    return tuple(runs) if runs else ((" ", 0),)


def count_luds(structures) -> Tuple[Dict[Tuple, Set], Dict[int, Set]]:
    """bpe_simulator.py:45-101: LUDS label -> structures that can spell it.  A structure without
    mixed tags belongs to the label obtained by merging equal neighbouring tags; a structure with a
    mixed tag (DM/TM/FM) belongs to every such label of the same total length that agrees with it
    character position by character position outside the mixed positions, and to not_parsed[length]."""
    converted: Dict[Tuple, Set] = {}
    mixed = []
    for st in structures:
        if any("M" in tag for tag, _ in st):
            mixed.append(st)
            continue
        label: List[Tuple[str, int]] = []
        for tag, ln in st:
            if label and label[-1][0] == tag:
                label[-1] = (tag, label[-1][1] + ln)
            else:
                label.append((tag, ln))
        converted.setdefault(tuple(label), set()).add(st)
    by_len: Dict[int, List[Tuple]] = {}
    for label in converted:
        by_len.setdefault(sum(ln for _, ln in label), []).append(label)
    spelled = {}
    not_parsed: Dict[int, Set] = {}
    for st in mixed:
        total = sum(ln for _, ln in st)
        tags = [tag for tag, ln in st for _ in range(ln)]
        for label in by_len.get(total, ()):
            if label not in spelled:
                spelled[label] = [tag for tag, ln in label for _ in range(ln)]
            if all(a == b or "M" in a or "M" in b for a, b in zip(spelled[label], tags)):
                converted[label].add(st)
        not_parsed.setdefault(total, set()).add(st)
    return converted, not_parsed


class BpePcfgSim(MonteCarlo):
    def __init__(self, model_path: str):
        grammars, terminals = read_bpe(model_path=model_path)
        self.grammars = expand_1d(grammars, minus_log_based=True)
        self.terminals = expand_2d(terminals, minus_log_based=True)
        self.converted, self.not_parsed = count_luds(grammars)
        self._flat = BpeFlat(self.grammars, self.terminals, self.converted, self.not_parsed, _CLASS)
        self._dev_model = None
        self._cls_lut: Dict[int, int] = {}

    # ---- sampling ------------------------------------------------------------------------------
    def sample1(self) -> Tuple[float, str]:
        seed, first = self._claim_ids(1)
        res = self._dev.sample(1, seed, first_id=first, max_draws=self.max_draws, sym_stride=self._flat.max_segments)
        self._dev.check()
        return float(res["ml2p"].cpu()[0]), self._strings_of(res, seed, first)[0]

    def _decode_rows(self, syms, lens) -> List[str]:
        return self._flat.decode_entries(syms, lens)

    def _sample_stride(self) -> int:
        return self._flat.max_segments

    # ---- scoring -------------------------------------------------------------------------------
    def calc_ml2p(self, pwd: str) -> Tuple[float, List[str]]:
        ml2p, parts = self.score_batch([pwd])
        return ml2p[0], parts[0]

    def score_batch(self, pwds: List[str]):
        if not pwds:
            return [], []
        cps, cls, off = self.encode(pwds)
        ml2p = self._dev.score_bpe(cps, cls, off).cpu().numpy().tolist()
        return ml2p, [[p] for p in pwds]

    def _score_unique_device(self, lines):
        """parse_file's scoring on lines the native ingest encoded: the device vector of -log2 p."""
        return self._dev.score_bpe(lines.cps, self._classes(lines.cps), lines.off)

    def encode(self, pwds: List[str]):
        """Passwords -> (code points, luds character classes, offsets) for mc_score_bpe."""
        cps, off, _ = self._flat.encode_text(pwds)
        return cps, self._classes(cps), off

    def _classes(self, cps: np.ndarray) -> np.ndarray:
        lut = self._cls_lut
        uniq = np.unique(cps)
        for c in uniq.tolist():
            if c not in lut:
                lut[c] = char_class(chr(c))
        table = np.array([lut[c] for c in uniq.tolist()], dtype=np.uint8)
        return table[np.searchsorted(uniq, cps)] if len(cps) else np.zeros(0, dtype=np.uint8)


def wrapper(model_path: str, testing_set, save2, size: int = 1000000):
    """bpe_simulator.py:160-172: the driver flow with a re-sort and the +1."""
    from . import _cli
    _cli.driver_flow(BpePcfgSim(model_path=model_path), testing_set, save2, size, need_resort=True, add1=True)


def main(argv=None):
    """The reference's command line (bpe_simulator.py:175-199)."""
    from . import _cli
    return _cli.bpe_simulator(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(main)
