#!/usr/bin/env python3
"""ISA audit for kernels that issue vector-memory loads from inline asm (hipcc neither counts them nor knows that their
destination registers are written later): for every such load, follow EVERY control-flow path until one of the
hand-placed waits (an `s_waitcnt vmcnt` inside an ASMSTART/ASMEND block, or the `ndcn-wait-begin` marker in front of a
jump table of them) and report any instruction that reads or
writes a destination register on the way (a copy, a spill, a re-use - all of them silent corruption).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s ndcn_amd/csrc/spmm_rec.hip
    python tools/audit_async_regs.py /tmp/k.s [kernel-name-substring]
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def audit(name, lines):
    # instruction list with asm-block membership
    ins, labels, in_asm = [], {}, False
    for l in lines:
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if in_asm and 'ndcn-wait-begin' in t:                 # a run-time-count wait: a jump table of s_waitcnt follows
            ins.append(('s_waitcnt vmcnt(rt)', True))
            continue
        t = t.split(';')[0].strip()
        if not t or (t.startswith('.') and not t.endswith(':')):
            continue
        if t.endswith(':'):
            labels[t[:-1]] = len(ins)
            continue
        ins.append((t, in_asm))
    n = len(ins)

    def succ(i):
        t = ins[i][0]
        op = t.split()[0]
        if op == 's_endpgm':
            return []
        if op == 's_branch':
            return [labels[t.split()[1]]]
        if op.startswith('s_cbranch'):
            return [labels[t.split()[1]], i + 1]
        return [i + 1] if i + 1 < n else []

    problems = 0
    loads = [i for i, (t, a) in enumerate(ins) if a and re.match(r'(global|buffer)_load_dword', t) and 'lds' not in t.split()[0]]
    for i in loads:
        dest = regs(ins[i][0].split(',')[0])
        seen, stack = set(), list(succ(i))
        while stack:
            j = stack.pop()
            if j in seen or j >= n:
                continue
            seen.add(j)
            t, a = ins[j]
            if a and t.startswith('s_waitcnt') and 'vmcnt' in t:
                continue                                  # path ends at a hand-placed wait
            if not t.startswith('s_') and (regs(t) & dest):
                # another asm load of the same family writing the same registers again is a re-request, also wrong
                problems += 1
                if problems <= 8:
                    print('  %s: "%s" touches v%s while "%s" is in flight' % (name[:60], t, sorted(regs(t) & dest)[:4], ins[i][0]))
                continue
            stack.extend(succ(j))
    return len(loads), problems


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    total = 0
    for m in re.finditer(r'^(_Z\w+):\s*; @\1', txt, re.M):
        name = m.group(1)
        if want not in name:
            continue
        end = txt.index('.Lfunc_end', m.end())                # a kernel may hold several s_endpgm
        nl, pr = audit(name, txt[m.end():end].split('\n'))
        if nl:
            print('%-90s asm loads %3d  problems %d' % (name[:90], nl, pr))
            total += pr
    print('TOTAL problems', total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
