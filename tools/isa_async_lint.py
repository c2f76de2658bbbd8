#!/usr/bin/env python3
"""Lint for gfx950 assembly (hipcc --save-temps *.s) of kernels that issue LDS reads from INLINE ASSEMBLY.

The compiler treats `asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))` as an instruction whose result exists
when the statement ends.  The hardware returns the data tens to hundreds of cycles later (in order with the other LDS
operations of the wave, counted by lgkmcnt).  Two things the compiler is therefore free to do, and the hardware punishes:

  * read-before-land:   an instruction reads a destination register of an in-flight read before an `s_waitcnt lgkmcnt`
                        has retired it (the kernels place those waits by hand);
  * clobber-before-land: the destination is DEAD in the source program (e.g. a look-ahead read whose value is replaced by a
                        later read), so the register allocator re-uses it as a scratch register while the read is still
                        in flight -- the returning data then overwrites the scratch value.  This is the root cause of
                        the round-3 "S2D one-plane split-K" wrong results: the stale look-ahead a0
                        read after a chunk's last tap landed in the register the compiler used for the next fragment address.

The scan is linear in program order per kernel (fall-through order; labels keep the state) -- a heuristic, but every
finding is a real hazard candidate to be read in the .s file.  Exit status 1 when anything is flagged.

usage: isa_async_lint.py file.s [--kernel SUBSTR] [--verbose]
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
ASYNC_DS = re.compile(r'^(ds_read\w*|ds_load\w*)\b')
WAIT = re.compile(r'^s_waitcnt\b(.*)')
LGKM = re.compile(r'lgkmcnt\((\d+)\)')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def split_ops(rest):
    ops, depth, cur = [], 0, ''
    for ch in rest:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            ops.append(cur.strip())
            cur = ''
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


NO_DST = ('ds_write', 'ds_store', 'buffer_store', 'global_store', 'flat_store', 'scratch_store', 's_', 'v_cmpx', ';')


def parse_blocks(lines):
    """Basic blocks: [(label or None, [(line_no, text, in_asm)], successors-by-label, falls_through)]"""
    blocks, cur, label = [], [], None
    in_asm = False

    def close(succ, fall):
        nonlocal cur, label
        blocks.append({'label': label, 'ins': cur, 'succ': succ, 'fall': fall})
        cur, label = [], None

    for no, raw in lines:
        s = raw.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.startswith(';'):
            continue
        m = re.match(r'^(\.LBB\w+):', s)
        if m:
            if cur or label is not None:
                close([], True)
            label = m.group(1)
            continue
        if s.startswith('.') or s.endswith(':'):
            continue
        s = s.split(';')[0].strip()
        if not s:
            continue
        for part in s.split('\n'):
            part = part.strip()
            if not part:
                continue
            mn = part.split(' ')[0]
            if mn == 's_branch':
                close([part.split()[-1]], False)
            elif mn.startswith('s_cbranch'):
                close([part.split()[-1]], True)
            elif mn == 's_endpgm':
                close([], False)
            else:
                cur.append((no, part, in_asm))
    if cur or label is not None:
        close([], False)
    return blocks


def run_block(ins, state, findings):
    inflight = list(state)
    for no, s, in_asm in ins:
        mn, _, rest = s.partition(' ')
        if WAIT.match(s):
            g = LGKM.search(s)
            if g is None:
                if 'vmcnt' in s or 'expcnt' in s:
                    continue
                n = 0
            else:
                n = int(g.group(1))
            inflight = inflight[len(inflight) - n:] if n and len(inflight) > n else ([] if n == 0 else inflight)
            continue
        if mn.startswith('s_'):
            if mn.startswith('s_load') or mn.startswith('s_buffer_load'):
                inflight.append((no, frozenset(), s))   # scalar loads share lgkmcnt (and may return out of order: the
            continue                                    # compiler waits lgkmcnt(0) for them, which retires everything)
        ops = split_ops(rest)
        is_ds = mn.startswith('ds_')
        is_read = bool(ASYNC_DS.match(mn))
        has_dst = is_read or (not is_ds and not mn.startswith(NO_DST))
        dst = regs(ops[0]) if (has_dst and ops) else set()
        src = set()
        for o in (ops[1:] if has_dst else ops):
            src |= regs(o)
        for (l0, d0, t0) in inflight:
            if src & d0:
                findings.add((no, 'read-before-land', s, l0, t0))
            if dst & d0 and not is_read:     # (LDS reads return in order: a later read into the same register is fine)
                findings.add((no, 'clobber-before-land', s, l0, t0))
        if is_ds:
            # only reads issued from inline asm are invisible to the compiler's own waitcnt insertion
            inflight.append((no, frozenset(dst) if (is_read and in_asm) else frozenset(), s))
    return tuple(inflight)


def lint_kernel(name, lines, verbose=False):
    blocks = parse_blocks(lines)
    by_label = {b['label']: i for i, b in enumerate(blocks) if b['label']}
    findings = set()
    seen = [set() for _ in blocks]
    work = [(0, ())]
    while work:
        i, st = work.pop()
        key = tuple((l0, d0) for (l0, d0, _) in st)
        if key in seen[i] or len(seen[i]) >= 12:
            continue
        seen[i].add(key)
        out = run_block(blocks[i]['ins'], st, findings)
        # the in-order count only matters together with registers: drop leading entries without registers
        while out and not out[0][1]:
            out = out[1:]
        for lab in blocks[i]['succ']:
            if lab in by_label:
                work.append((by_label[lab], out))
        if blocks[i]['fall'] and i + 1 < len(blocks):
            work.append((i + 1, out))
    return sorted(findings)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    verbose = '--verbose' in sys.argv
    sub = None
    if '--kernel' in sys.argv:
        sub = sys.argv[sys.argv.index('--kernel') + 1]
        args = [a for a in args if a != sub]
    bad = 0
    for path in args:
        cur, name, kernels = None, None, []
        with open(path) as f:
            for no, line in enumerate(f, 1):
                m = re.match(r'^(_Z\w+):', line)
                if m:
                    name = m.group(1)
                    cur = []
                    kernels.append((name, cur))
                    continue
                if cur is not None:
                    if '.end_amdhsa_kernel' in line or line.startswith('.Lfunc_end'):
                        cur = None
                        continue
                    cur.append((no, line))
        for name, lines in kernels:
            if sub and sub not in name:
                continue
            f = lint_kernel(name, lines, verbose)
            uses_asm = any(';;#ASMSTART' in l for _, l in lines)
            if not uses_asm:
                continue
            print('%s: %d finding(s)' % (name, len(f)))
            for (no, kind, s, l0, t0) in f[:40 if not verbose else None]:
                print('  line %d: %s: `%s`  <- in-flight since line %d: `%s`' % (no, kind, s, l0, t0))
            bad += len(f)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
