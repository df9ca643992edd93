"""Register / spill table of the kernels in a hipcc -save-temps device assembly file.  usage: python tools/probe/regs.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
md = s[s.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    if flt in name:
        print('%-90s vgpr %s spill %s sgpr-spill %s lds %s scratch %s' % (name[-90:], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
