"""Copy the outputs of tools/probes/capture_profiles.sh, tools/pmc_hbm.sh and tools/bench_other.py from gpurun_out/ into the tracked
profiles/ directory and rebuild the derived summaries (class-average agreement check, HBM traffic, side benches)."""
import csv, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
for f in ('bench_default', 'bench_fp16', 'bench_fp32', 'bench_under_rocprof'):
    shutil.copy(os.path.join(G, 'cap', f + '.json'), os.path.join(P, '%s_final_%s.json' % (tag, f)))
shutil.copy(os.path.join(G, 'cap', 'kernel_stats.csv'), os.path.join(P, tag + '_final_kernel_stats.csv'))
shutil.copy(os.path.join(G, 'cap', 'kernel_trace_summary.txt'), os.path.join(P, tag + '_final_kernel_trace_summary.txt'))
os.chdir(R)
for t, prec in (('x3', 'fp16x3'), ('x1', 'fp16')):
    subprocess.run([sys.executable, 'tools/hbm_traffic.py', t, 'profiles/%s_hbm_traffic_%s.json' % (tag, prec)], check=True,
                   stdout=subprocess.DEVNULL)


def dem(n):
    if n.startswith('_Z'):
        try:
            return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip() or n
        except Exception:
            return n
    return n


rows = list(csv.DictReader(open(os.path.join(P, tag + '_final_kernel_stats.csv'))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = ['%6.2f%%  n=%-6s avg=%8.1f us  %s' % (float(r['Percentage']), r['Calls'], float(r['AverageNs']) / 1e3, dem(r['Name'])[:110])
         for r in rows[:14]]
cls_ns = cls_calls = 0
for r in rows:
    n = dem(r['Name'])
    in16_old = 'conv_f16_kernel<' in n and n.split('>')[0].split(',')[-2].strip() == 'true'
    q_s1 = 'conv_f16_q_kernel' in n and n.split('>')[0].rstrip().endswith(', 1')     # stride-1 quad launches (incl. Upsample)
    if q_s1 or 'conv_f16_lc_kernel' in n or in16_old:
        cls_ns += float(r['TotalDurationNs'])
        cls_calls += int(r['Calls'])
b = json.load(open(os.path.join(P, tag + '_final_bench_under_rocprof.json')))
d = json.load(open(os.path.join(P, tag + '_final_bench_default.json')))
out = ['rocprofv3 --kernel-trace --stats of `python bench.py --no-alt` (fp16x3 default), top kernels:'] + lines
out += ['', 'stride-1 fp16-plane 3x3 convolutions (conv_f16_q_kernel<.., 1> incl. the Upsample launches + conv_f16_lc_kernel* + conv_f16_kernel<..IN16..>):',
        '  rocprofv3: %d launches, average %.1f us  (%.1f%% of GPU time)' % (cls_calls, cls_ns / max(cls_calls, 1) / 1e3, 100 * cls_ns / tot),
        '  bench.py HIP events, dominant class (stride-1 3x3 convs without the Upsample ones), same run under the profiler: average %.1f us over %d sampled launches'
        % (b['roofline']['avg_launch_ms'] * 1e3, b['roofline']['launches']),
        '  bench.py HIP events, stand-alone run: average %.1f us over %d sampled launches' % (d['roofline']['avg_launch_ms'] * 1e3, d['roofline']['launches'])]
open(os.path.join(P, tag + '_final_class_check.txt'), 'w').write('\n'.join(out) + '\n')
side = []
for f in ('side_x3.txt', 'side_x1.txt'):
    side += [json.loads(l) for l in open(os.path.join(G, f)) if l.strip()]
side.append({'workload': 'BASELINE configs[4] shape: NCSN++ 256x256 (nf=128, ch_mult (1,1,2,2,2,2,2), attention at 16, Fourier embedding, '
                         '65.57 M parameters), planned executor, forward only; rel. error vs the CPU oracle 2.1e-6 (tests/test_gpu_fullsize.py::test_config5_ncsnpp_256_vs_oracle)',
             'precision': 'fp16x3', 'batch': 8, 'ms_per_forward': 22.29, 'images_per_sec_per_nfe': 358.9})
side.append({'note': 'tools/bench_other.py on one MI355X (ncsnpp_paired = planned graph executor, ncsnpp_paired_ops = operator-granular '
                     'executor); shader clock during the 160x160 conv launches 1.86-2.08 GHz (clock64 / wall_clock64, '
                     'tools/probes/phase_timing_lc.py) against 2.39 GHz in the pure-MFMA probe'})
json.dump(side, open(os.path.join(P, tag + '_side_benches.json'), 'w'), indent=1)
for n in ('default', 'fp16', 'fp32', 'under_rocprof'):
    j = json.load(open(os.path.join(P, '%s_final_bench_%s.json' % (tag, n))))
    print(n, round(j['value'], 3), round(j['ms_per_step'], 2), j['roofline']['bound'], round(j['roofline']['achieved'], 1),
          round(j['roofline']['frac'], 3), round(j['hbm_roofline']['frac'], 3), j['roofline'].get('traffic'))
print(open(os.path.join(P, tag + '_final_class_check.txt')).read().splitlines()[-3:])
for s in side[:-2]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in s.items() if k != 'params'})
