"""print a one-line digest of a bench.py JSON line read from stdin"""
import json, sys
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r = d['roofline']
print(sys.argv[1] if len(sys.argv) > 1 else '', 'value %.3f steps/s  %.1f ms/step  3task %.2f/s  drop0.1 %.2f/s  conv %.2f ms/pass  top %s %.1f TF' % (
    d['value'], d['ms_per_step'], d.get('configs1_3task', {}).get('value', 0), d.get('dropout_0.1', {}).get('value', 0), r['conv_stack']['ms_per_pass'], r['kernel'], r['achieved']))
