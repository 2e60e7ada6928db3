"""(Record of a measured and dropped experiment: DESIGN.md section 3.1, "dataflow form".  The kernel is no longer in
ver_sca.hip; to rebuild it, paste df_kernel.tmpl in front of k_sca_bwd and restore the VER_SCA_FWD_DF dispatch.)
Regenerates k_sca_fwd_df in ver_sca.hip: the iteration body (phase A / phase B) is the text of k_sca_fwd_cs's loop
body with the pair-range expressions of the dataflow deal substituted, so the two kernels cannot drift apart."""
import re, sys
p = 'vln-ver_amd/csrc/ver_sca.hip'
s = open(p).read()
L = s.split('\n')
i0 = [i for i, l in enumerate(L) if '// ---------------- phase A: lane = sampling point l8 of voxel (s4, v2)' in l][0]
i1 = [i for i, l in enumerate(L) if 'n0 = n1; n1 = n2; n2 = n3;' in l and i > i0][0]
body = [l for l in L[i0:i1] if 'VER_TL(' not in l]
txt = '\n'.join(body)
txt = txt.replace('p_hi - p_lo - 4 * it', 'TP - pbase').replace('p_lo + 4 * it + j', 'pbase + j')
assert 'p_lo' not in txt and 'p_hi' not in txt
txt = '\n'.join(l[4:] if l.startswith('    ') else l for l in txt.split('\n'))
kernel = open('scratch/r04/df_kernel.tmpl').read().replace('@@BODY@@', txt)
beg = s.index('// ------------------------------------------------------------------------------------------\n// Forward, DATAFLOW form')
end = s.index('// ------------------------------------------------------------------------------------------\ntemplate <int HD, int G, int P, typename VT>\n__global__ __launch_bounds__(512) void k_sca_bwd(')
s = s[:beg] + kernel.rstrip('\n') + '\n\n' + s[end:]
open(p, 'w').write(s)
print('ok', len(txt.split('\n')), 'body lines')
