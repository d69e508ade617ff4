"""Static check of hipcc output for one miscompilation pattern seen on gfx950 (ROCm 7.2): register-allocator copies
(v_accvgpr_write / v_accvgpr_read / scratch spills / VGPR moves) placed in the exit block of an exec-masked (divergent) loop or if-region BEFORE the `s_or_b64 exec,
exec, s[..]` that restores the lane mask.  Those vector instructions run with a partial (often empty) EXEC, so the "saved" lanes
keep stale register contents; once the value is read back under the full mask the kernel computes with garbage addresses.

usage: check_exec_restore.py file.s [file.s ...]   (exit code 1 if any kernel shows the pattern)
Returns a list of (kernel, line number, instruction) from scan(path).
"""
import re
import sys

# register-allocator traffic: VGPR <-> AGPR copies, scratch spills, plain VGPR-to-VGPR moves
VEC = re.compile(r"^\s*(v_accvgpr_(write|read)_b32 |scratch_(store|load)|buffer_(store|load)_dword\S* v\S+, off, s\[\d+:\d+\], 0|v_mov_b32_e32 v\d+, v\d+\s*$)")
BENIGN = re.compile(r"^\s*v_(readlane|readfirstlane|writelane)_b32")      # ignore EXEC by definition
RESTORE = re.compile(r"^\s*s_or_b64 exec, exec, ")
LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*; @")


def scan(path):
    hits = []
    kernel = None
    block = []          # (lineno, text) of vector instructions seen since the last label
    with open(path) as fh:
        for n, line in enumerate(fh, 1):
            m = FUNC.match(line)
            if m:
                kernel = m.group(1)
            if LABEL.match(line):
                block = []
                continue
            t = line.split(";")[0].rstrip()
            if not t.strip():
                continue
            if RESTORE.match(t):
                for ln, txt in block:
                    hits.append((kernel, ln, txt.strip()))
                block = []
                continue
            if VEC.match(t) and not BENIGN.match(t):
                block.append((n, t))
            elif re.match(r"^\s*s_(cbranch|branch|endpgm|barrier)", t):
                block = []          # control leaves the block: later restores belong to other paths
    return hits


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for k, ln, txt in scan(p):
            print(f"{p}:{ln}: [{k}] vector instruction ahead of the exec restore of its block: {txt}")
            bad += 1
    sys.exit(1 if bad else 0)
