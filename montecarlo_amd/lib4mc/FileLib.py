"""Line counting helper with the side effect parse_file relies on: the file is
rewound before and after (reference lib4mc/FileLib.py:5-27; it is what lets the
secondary-training loop re-read the same test file every round)."""
import sys
from typing import TextIO

_CHUNK = 8 << 20


def wc_l(file: TextIO, new_line: str = "\n", silence: bool = False) -> int:
    can_seek = file.seekable()
    if can_seek:
        file.seek(0)
    elif not silence:
        print("WARNING: file cannot seekable", file=sys.stderr)
    lines = 1                      # the reference counts separators + 1
    for block in iter(lambda: file.read(_CHUNK), ""):
        lines += block.count(new_line)
    if can_seek:
        file.seek(0)
    return lines
