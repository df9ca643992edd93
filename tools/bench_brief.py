"""print a one-line digest of bench.py JSON lines: files given as arguments (label = file name), or one line on stdin when piped"""
import json, sys


def brief(label, text):
    d = json.loads([l for l in text.splitlines() if l.startswith('{')][-1])
    r = d['roofline']
    print(label, 'value %.3f steps/s  %.1f ms/step  3task %.2f/s  drop0.1 %.2f/s  conv %.2f ms/pass  top %s %.1f TF' % (
        d['value'], d['ms_per_step'], d.get('configs1_3task', {}).get('value', 0), d.get('dropout_0.1', {}).get('value', 0),
        r['conv_stack']['ms_per_pass'], r['kernel'], r['achieved']))


files = [a for a in sys.argv[1:] if a.endswith('.json') or a.endswith('.log')]
if files:
    for f in files:
        brief(f, open(f).read())
elif not sys.stdin.isatty():
    brief(sys.argv[1] if len(sys.argv) > 1 else '', sys.stdin.read())
else:
    sys.exit('usage: bench_brief.py FILE.json ... | bench_brief.py [label] < line')
