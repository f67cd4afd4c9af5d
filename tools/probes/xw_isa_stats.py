"""static look at conv_xw's generated code: instructions per MFMA gap, spill traffic, instruction mix (per kernel instantiation)"""
import collections, re, sys
txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else 'ILi3ELb1ELb1E'
start = [i for i, l in enumerate(txt) if re.match(r'^_ZN3csd14conv_x\w_kernel' + want + r'\w*:', l)][0]
end = [i for i in range(start, len(txt)) if 's_endpgm' in txt[i]][0]
body = [l.strip() for l in txt[start:end] if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
print('instructions', len(body))
c = collections.Counter(l.split()[0] for l in body)
print(' '.join('%s:%d' % kv for kv in c.most_common(40)))
gaps, cur, kinds = [], [], collections.Counter()
for l in body:
    if l.startswith('v_mfma'):
        gaps.append(cur); cur = []
    else:
        cur.append(l.split()[0])
inner = [g for g in gaps if len(g) < 40]
print('MFMAs %d; gaps < 40 instr: %d, mean %.2f instr per gap' % (len(gaps), len(inner), sum(map(len, inner)) / len(inner)))
print('gap size histogram', sorted(collections.Counter(len(g) for g in inner).items()))
k = collections.Counter(i for g in inner for i in g)
print('in-gap mix:', ' '.join('%s:%d' % kv for kv in k.most_common(30)))
