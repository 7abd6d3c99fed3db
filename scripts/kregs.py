"""CPU: register / LDS / scratch use per kernel from a -save-temps assembly file (vistaocr_amd/csrc/build/*.s)."""
import re, sys
s = open(sys.argv[1]).read()
for blk in s.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name[:70], 'agpr', blk.split('\n')[0].strip(), 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'spill', g('vgpr_spill_count'),
          'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
