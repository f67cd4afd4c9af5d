"""Turn the two PMC summaries of tools/pmc_hbm.sh (FETCH_SIZE, WRITE_SIZE per (kernel, grid), one PC step = 2 network
evaluations) into profiles/<tag>_hbm_traffic.json.  Calibration on this run's own streaming kernels
(langevin_update / scale_rows / assemble_input, whose byte counts are known exactly): WRITE_SIZE is exact in KiB,
FETCH_SIZE reports exactly half of the bytes read (MI355X_MICROARCH.md, HBM section) -> fetch = 2 * FETCH_SIZE KiB."""
import json, re, sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r'(.+?)\s+wgs=(\d+)\s+(\w+)\s+avg=([\d.e+]+) \(n=(\d+)\)', line.strip())
        if m:
            out[(m.group(1).strip(), int(m.group(2)))] = (float(m.group(4)), int(m.group(5)))
    return out


def main(tag, outpath):
    f = parse('gpurun_out/pmc_%s_FETCH_SIZE.txt' % tag)
    w = parse('gpurun_out/pmc_%s_WRITE_SIZE.txt' % tag)
    rows, cls = [], {'launches': 0, 'fetch': 0.0, 'write': 0.0}
    any_xk = any(k[0].startswith('conv_xk_kernel') for k in f)
    for key in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
        if key not in w:
            continue
        fetch = 2.0 * f[key][0] * 1024
        write = w[key][0] * 1024
        n = f[key][1]
        rows.append({'kernel': key[0], 'workgroups': key[1], 'launches_in_run': n,
                     'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write})
        name = key[0]
        in16_old = name.startswith('conv_f16_kernel') and name.rstrip('>').split(',')[-2].strip() == 'true'
        q_resample = False
        if name.startswith('conv_f16_q_kernel'):      # <MQ, NS, MASK, PWC, S, NTQ, F8, UP4>: stride-2 / phase-decomposed Upsample = the resample class
            targs = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')]
            q_resample = targs[4] != '1' or (len(targs) > 7 and targs[7] == 'true')
        # the class the bench's roofline object prices: in a run whose big 3x3 layers are on conv_xk (fp16x3) that kernel's launches
        # alone (profiler class 'conv3x3'); otherwise every 3x3 stride-1 kernel of the mode
        if any_xk:
            in_class = name.startswith('conv_xk_kernel')
        else:
            in_class = (name.startswith(('conv_ff_kernel', 'conv_fx_kernel', 'conv_f16_q_kernel', 'conv_f16_lc_kernel')) or in16_old) and not q_resample
        if in_class:
            cls['launches'] += n
            cls['fetch'] += fetch * n
            cls['write'] += write * n
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0`, '
                     'tools/pmc_hbm.sh; fetch = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB (calibrated on the sampler '
                     'update kernels of the same run)',
           'class_definition': 'conv_xk_kernel launches (profiler class conv3x3)' if any_xk else 'every 3x3 stride-1 kernel of the mode',
           'conv3x3_class': {'launches_in_run': cls['launches'], 'run': 'bench.py --steps 1 --warmup 0 = 1 timed + 1 profiled PC step',
                             'hbm_bytes_per_launch': (cls['fetch'] + cls['write']) / max(cls['launches'], 1),
                             'fetch_bytes_per_launch': cls['fetch'] / max(cls['launches'], 1),
                             'write_bytes_per_launch': cls['write'] / max(cls['launches'], 1)},
           'kernels': rows[:40]}
    json.dump(res, open(outpath, 'w'), indent=1)
    print(json.dumps(res['conv3x3_class']))
    for r in rows[:6]:
        print(r)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
