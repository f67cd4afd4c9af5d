"""Build-time check of conv_xp.hip's generated code (csrc/Makefile target check-xp, run by __graft_entry__.build()).

The kernel issues its matrix instructions as asm statements, so hipcc inserts none of the wait states an accumulator access needs
(MI355X: no hardware interlock between a matrix write and a vector read of the same register).  The source is structured so that
hipcc never has a reason to touch an accumulator; this script proves it on the ISA of every conv_xp_kernel instantiation:
  * the matrix instructions use exactly 2 NT accumulator tuples, the same registers throughout the kernel;
  * no v_accvgpr_mov / v_accvgpr_write (or any other non-matrix instruction) writes an accumulator register;
  * every read of an accumulator register sits behind the epilogue's tied wait (s_nop 15) with no matrix instruction in between.
usage: check_xp_isa.py conv_xp.s"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'a\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'a(\d+)', tok)
    return {int(m.group(1))} if m else set()


def check(name, lines):
    acc, tuples = set(), set()
    for ln in lines:
        if ln.startswith('v_mfma'):
            dst = ln.split()[1].rstrip(',')
            tuples.add(dst)
            acc |= regs(dst)
    nt = int(re.search(r'conv_xp_kernelILi(\d)E', name).group(1))
    errs = []
    if len(tuples) != 2 * nt or len(acc) != 32 * nt:
        errs.append('%d accumulator tuples (%d registers), expected %d (%d)' % (len(tuples), len(acc), 2 * nt, 32 * nt))
    behind_tie = False
    for i, ln in enumerate(lines):
        if ln.startswith('v_mfma'):
            behind_tie = False
            continue
        if ln.startswith('s_nop 15'):
            behind_tie = True
            continue
        toks = [t.rstrip(',') for t in ln.split()[1:]]
        touched = [t for t in toks if regs(t) & acc]
        if not touched:
            continue
        writes = bool(toks) and bool(regs(toks[0]) & acc) and not ln.startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store'))
        if writes:
            errs.append('line %d writes an accumulator: %s' % (i, ln))
        elif not behind_tie:
            errs.append('line %d reads an accumulator in the shadow of a matrix instruction: %s' % (i, ln))
    return nt, len(tuples), errs


def main(path):
    text = open(path).read().split('\n')
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r'^(_ZN3csd14conv_xp_kernel\w+):', ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            t = ln.strip()
            if t.startswith('s_endpgm'):
                kernels[name] = cur
                cur = None
            elif t and not t.startswith((';', '.')):
                cur.append(t)
    if not kernels:
        print('check_xp_isa: no conv_xp_kernel in', path)
        return 1
    bad = 0
    for name, lines in sorted(kernels.items()):
        nt, ntup, errs = check(name, lines)
        print('%s: NT = %d, %d accumulator tuples, %d instructions: %s' % (name, nt, ntup, len(lines), 'ok' if not errs else 'FAILED'))
        for e in errs[:10]:
            print('   ', e)
        bad += bool(errs)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
