"""Audit of a hand-placed inline-asm kernel: everything hipcc emitted by itself inside the loops.
    hipcc ... -S --cuda-device-only unit.hip -o unit.s ; python tools/asm_audit.py unit.s <kernel-name-substring>
Every instruction of the software-pipelined kernels' loops is an asm statement (between ;;#ASMSTART / ;;#ASMEND); whatever
appears outside those markers inside a loop block is compiler code: register copies (v_mov / v_accvgpr) of values that asm
loads may still be landing in are BUGS, scratch accesses are spills, s_waitcnt are drains hipcc added on its own."""
import re
import sys
from collections import Counter


def audit(path, key, verbose=True):
    s = open(path).read()
    fns = re.split(r'\n(?=_ZN5nmfmu\w+:)', s)
    out = {}
    for fn in fns[1:]:
        name = fn.split(':')[0]
        if key not in name:
            continue
        in_loop, in_asm = False, False
        comp, asm_ops = Counter(), Counter()
        comp_lines = []
        # only the loops that hold MFMAs (the tile loop), not the epilogue's: pass 1 finds their headers
        hdr_of, cur, mfma_hdrs = {}, None, set()
        for line in fn.split('\n'):
            m = re.match(r'(\.LBB\d+_\d+):', line.strip())
            if m:
                h = re.search(r'Header=(BB\d+_\d+)', line)
                cur = ('.L' + h.group(1)) if h else (m.group(1) if 'Loop Header' in line else None)
            elif cur and 'v_mfma' in line:
                mfma_hdrs.add(cur)
        cur = None
        for line in fn.split('\n'):
            t = line.strip()
            m = re.match(r'(\.LBB\d+_\d+):', t)
            if m:
                h = re.search(r'Header=(BB\d+_\d+)', line)
                cur = ('.L' + h.group(1)) if h else (m.group(1) if 'Loop Header' in line else None)
                in_loop = cur in mfma_hdrs
                continue
            if t.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            if not in_loop:
                continue
            op = t.split()[0]
            if in_asm:
                asm_ops[op] += 1
            else:
                comp[op] += 1
                comp_lines.append(t)
        out[name] = (asm_ops, comp, comp_lines)
        if verbose:
            print(name)
            print('  asm instructions in loops :', dict(asm_ops.most_common()))
            print('  COMPILER instructions in loops:', dict(comp.most_common()))
            bad = [l for l in comp_lines if re.match(r'(v_mov|v_accvgpr|scratch_|buffer_|s_waitcnt|v_readlane|v_writelane)', l)]
            for l in bad[:40]:
                print('    !!', l)
    return out


if __name__ == '__main__':
    audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'sp_kernel')
