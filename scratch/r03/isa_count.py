"""Instruction census of one kernel in a hipcc -S listing: python isa_count.py file.s <substring of the mangled name> ..."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
for pat in sys.argv[2:]:
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m and pat in m.group(1):
            j = i
            while '.amdhsa_kernel' not in lines[j]:
                j += 1
            body = '\n'.join(lines[i:j])
            c = lambda p: len(re.findall(p, body))
            vg = re.search(r'\.set %s\.num_vgpr, (\d+)' % re.escape(m.group(1)), '\n'.join(lines[j:j + 80]))
            print(m.group(1)[4:34], 'vgpr', vg.group(1) if vg else '?', 'pk_fma_f16', c('v_pk_fma_f16'), 'fma_mix', c('v_fma_mix'),
                  'pk_fma_f32', c('v_pk_fma_f32'), 'fma_f32', c(r'v_fma_f32|v_fmac_f32'), 'cvt_f32_f16', c('v_cvt_f32_f16'),
                  'and', c('v_and_b32'), 'lshl', c('v_lshlrev_b32'), 'scratch', c('scratch_'), 'VALU', c(r'\n\tv_'),
                  'SALU', c(r'\n\ts_'), 'DS', c(r'\n\tds_'), 'nop', c('s_nop'))
