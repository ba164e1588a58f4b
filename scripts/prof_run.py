#!/usr/bin/env python3
"""rocprofv3 passes over one command, per-kernel summaries on stdout (run on the GPU box).

    python scripts/prof_run.py TAG [--match SUBSTR] [--pmc "A B" --pmc "C"] [--no-stats] -- <command ...>

One `--kernel-trace --stats` pass (unless --no-stats) and one `--pmc` pass per group, each in its own
run (gpurun refuses counter passes combined with other trace domains); outputs under
gpurun_out/<TAG>/; prints the kernel-stats head and per-kernel counter means for kernels whose name
contains SUBSTR.  The summaries to be judged are copied into profiles/ by hand.
"""
import os, sys, csv, glob, subprocess, collections, argparse

ap = argparse.ArgumentParser()
ap.add_argument('tag')
ap.add_argument('--match', default='spmm')
ap.add_argument('--pmc', action='append', default=[])
ap.add_argument('--no-stats', action='store_true')
argv = sys.argv[1:]
split = argv.index('--') if '--' in argv else len(argv)
a = ap.parse_args(argv[:split])
cmd = argv[split + 1:]
root = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(root, 'gpurun_out', a.tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
cmd = [c if not (c.endswith('.py') and not os.path.isabs(c)) else os.path.join(root, c) for c in cmd]


def run(extra, sub):
    d = os.path.join(out, sub)
    full = ['rocprofv3'] + extra + ['--output-format', 'csv', '-d', d, '-o', 'run', '--'] + cmd
    with open(os.path.join(out, sub + '.log'), 'w') as f:
        subprocess.run(full, cwd='/tmp', env=env, stdout=f, stderr=subprocess.STDOUT)
    return d


if not a.no_stats:
    d = run(['--kernel-trace', '--stats'], 'stats')
    fs = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)
    if fs:
        print('== kernel stats (%s)' % fs[0])
        for i, line in enumerate(open(fs[0])):
            if i < 14:
                print(line.rstrip()[:220])
        subprocess.run(['cp', fs[0], os.path.join(out, 'kernel_stats.csv')])
    log = open(os.path.join(out, 'stats.log')).read().splitlines()
    print('\n'.join(l[:300] for l in log if not l.startswith('W2') and not l.startswith('E2'))[-3000:])
for i, grp in enumerate(a.pmc):
    d = run(['--pmc'] + grp.split() + ['--kernel-trace'], 'pmc%d' % i)
    fs = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    print('== pmc %s' % grp)
    if not fs:
        print('   no counter file; log tail:')
        print('\n'.join(open(os.path.join(out, 'pmc%d.log' % i)).read().splitlines()[-5:]))
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = (r['Kernel_Name'][:70], r['Counter_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    with open(os.path.join(out, 'pmc_summary.txt'), 'a') as sf:
        for (kn, cn), (cnt, tot) in sorted(agg.items()):
            if a.match in kn:
                line = '%-72s %-26s dispatches=%d mean=%.1f' % (kn, cn, cnt, tot / cnt)
                print(line)
                sf.write(line + '\n')
